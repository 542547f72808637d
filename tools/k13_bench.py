"""BASELINE cfg4's scorer alone: 1 x M get_close passes over 8 000 x 20 kb sequences at k = 13, 64-bit counts, sparse lists -- the leg
`secondary[3]` of bench.py without the rest (for A/B runs: MSC_NO_RANKS_HASH=1 keeps the merge kernel k_pair_sparse_mp).
    python tools/k13_bench.py [n] [length] [passes]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from meshclust2_amd import api, synth

def main():
    n4 = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
    len4 = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    passes = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    k4 = 13
    ctx = api.Context(0)
    ctx.enable_timing(True) if hasattr(ctx, "enable_timing") else None
    codes, _ = synth.family_codes(20260004, n4, len4)
    s4 = api.HistogramSet(ctx, k4, 64, n4, sparse_entries=n4 * (len4 + 64))
    for done in range(0, n4, 2000):
        b = synth.pack_batch(codes[done:done + 2000])
        s4.build_packed(done, len(codes[done:done + 2000]), b["packed"], b["n_bases"], b["seg_seq"], b["seg_start"], b["seg_end"], b["eff_len"], b["one_mers"])
    wtext = open(os.path.join(ROOT, "tests", "golden", "weights_k9_u32.txt")).read().replace("k: 9", "k: %d" % k4).replace("uint32_t", "uint64_t")
    trn = api.Trainer(ctx, api.Feature.from_text(ctx, wtext, 0), 0.9)
    res = []
    for j in range(3):
        res.append(trn.get_close(s4, None, s4, (j * 7919 + 1) % n4, m=n4))
    ctx.synchronize()
    ms = []
    closed = 0
    t0 = time.perf_counter()
    for j in range(passes):
        r = trn.get_close(s4, None, s4, (j * 7919 + 3) % n4, m=n4)
        closed += int(np.count_nonzero(r[0]))
        ms.append(ctx.last_kernel_ms()[0])
    ctx.synchronize()
    dt = time.perf_counter() - t0
    ent = float(np.mean([s4.entries(i) for i in range(0, n4, 16)]))
    print("k=13: %d x %d bp, %d passes: %.3f ms per pass wall, streaming kernel %.3f ms (%s); closed %d; %.0f stored bins per list: %.2f TB/s on 8 bytes per stored bin"
          % (n4, len4, passes, dt / passes * 1e3, float(np.mean(ms)), ctx.last_kernel_info()[0], closed, ent, (n4 + 1) * ent * 8 / (float(np.mean(ms)) * 1e-3) / 1e12))

main()
