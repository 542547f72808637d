#!/usr/bin/env python3
"""pairs/s of msc_score_multi (Q x M pass) vs Q single passes -- run on the GPU box."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshclust2_amd import api, synth
ctx = api.Context(0)
k, dt, n = 9, int(os.environ.get("MSC_SWEEP_DT", "32")), int(sys.argv[1]) if len(sys.argv) > 1 else 16384
codes = [synth.member(5, t // 20, t % 20, synth.template(5, t // 20, 1000)) for t in range(4000)]
hs = api.HistogramSet(ctx, k, dt, n)
b = synth.pack_batch(codes)
done = 0
while done < n:
    m = min(len(codes), n - done)
    if m < len(codes):
        b = synth.pack_batch(codes[:m])
    hs.build_packed(done, m, b["packed"], b["n_bases"], b["seg_seq"], b["seg_start"], b["seg_end"], b["eff_len"], b["one_mers"])
    done += m
golden = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
if os.environ.get("MSC_SWEEP_NOEMD"):       # a reference-trained model without the earth mover's distance (euclidean, normalized_vectors, simratio)
    blk = json.load(open(os.path.join(golden, "train_k7_u8_slow.json")))["block"]
    feat = api.Feature.from_text(ctx, "k: %d\nmode: 1\nmax_features: 3\nID: 0.8\nDatatype: uint%d_t\nfeature_set: 0\n" % (k, dt) + blk, 0)
else:
    feat = api.Feature.from_text(ctx, open(os.path.join(golden, "weights_k9_u32.txt")).read(), 0)
for nq in [int(x) for x in os.environ.get("MSC_SWEEP_NQ", "1,2,4,8,16").split(",")]:
    qs = np.arange(nq, dtype=np.uint32) * 3
    ts = []
    for it in range(4):
        t0 = time.perf_counter()
        api.score_multi(ctx, feat, hs, None, hs, qs, m=n)
        ts.append((time.perf_counter() - t0, ctx.last_kernel_ms()[0]))
    wall = np.median([a for a, _ in ts[1:]]); tiles = np.median([b_ for _, b_ in ts[1:]])
    print(json.dumps({"n_q": nq, "m": n, "wall_ms": round(wall * 1e3, 2), "tiles_ms": round(float(tiles), 3), "pairs_per_s_wall": round(nq * n / wall),
                      "pairs_per_s_kernel": round(nq * n / tiles * 1e3), "cand_GBps": round(n * 4 ** k * dt / 8 / tiles / 1e6, 1), "kernel": ctx.last_kernel_info()[0]}), flush=True)
