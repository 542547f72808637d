"""The long-list window pass (k_pair_ranks_items and the kernels around it) on a fixed list of passes: a cfg5-shaped set, the queries
drawn once, every pass the same whatever the kernels return -- for A/B runs under rocprofv3 (tools/items_bench.sh).
    python tools/items_bench.py [n_templates] [passes] [min_query_len]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from meshclust2_amd import api, synth

def main():
    n_t = int(sys.argv[1]) if len(sys.argv) > 1 else 800
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    min_len = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
    seed, per = 20260005, 5
    seqs = []
    for t in range(n_t):
        u = synth._unit(synth._stream(seed, 2 * t + 1, 2))
        length = int(round(np.exp(np.log(500.0) + float(u[0]) * (np.log(50000.0) - np.log(500.0)))))
        tmpl = synth.template(seed, t, length)
        for j in range(per):
            rate = 0.03 + 0.07 * j
            seqs.append(synth.to_ascii(synth.member(seed, t, j, tmpl, sub_rate=rate * 0.85, indel_rate=rate * 0.15)))
    n = len(seqs)
    lens = np.array([len(s_) for s_ in seqs])
    ctx = api.Context(0)
    hs = api.HistogramSet(ctx, 9, 16, n, sparse_entries=int(lens.sum()) + 4096)
    for off in range(0, n, 256):
        hs.build(seqs[off:off + 256], first_slot=off)
    wtext = open(os.path.join(ROOT, "tests", "golden", "weights_cfg5_k9.txt")).read().replace("uint8_t", "uint16_t")
    trn = api.Trainer(ctx, api.Feature.from_text(ctx, wtext, 0), 0.6)
    order = np.argsort(lens, kind="stable").astype(np.uint32)
    sl = lens[order]
    win = api.Window(ctx, hs, order)
    rng = np.random.default_rng(5)
    cand = [int(x) for x in rng.permutation(n) if lens[x] >= min_len][:passes]
    def one(q):
        a, b = int(np.searchsorted(sl, int(lens[q] * 0.6), "left")), int(np.searchsorted(sl, int(lens[q] / 0.6), "right"))
        closed = win.get_close(trn, a, b, hs, q)[0]
        return b - a, int(sl[a:b].sum()), len(closed)
    for q in cand[:5]:
        one(q)
    ctx.synchronize()
    pairs = kmers = ncl = 0
    t0 = time.perf_counter()
    for q in cand:
        p, km, cl = one(q)
        pairs += p; kmers += km; ncl += cl
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print("sequences %d passes %d candidates/pass %.0f k-mers/pass %.3g closed %d | %.1f us per pass wall, %.2f TB/s on 4 bytes per k-mer | last kernel %s"
          % (n, len(cand), pairs / len(cand), kmers / len(cand), ncl, dt / len(cand) * 1e6, 4 * kmers / dt / 1e12, ctx.last_kernel_info()[0]))

main()
