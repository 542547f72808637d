#!/usr/bin/env python3
"""pairs/s of the 1 x M pass with and without the divergence statistics (--feat slow) -- run on the GPU box."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshclust2_amd import api, synth
ctx = api.Context(0)
k, dt, n = 9, int(os.environ.get("MSC_SWEEP_DT", "32")), int(sys.argv[1]) if len(sys.argv) > 1 else 16384
codes = [synth.member(5, t // 20, t % 20, synth.template(5, t // 20, 1000)) for t in range(4000)]
hs = api.HistogramSet(ctx, k, dt, n)
b = synth.pack_batch(codes)
for done in range(0, n, len(codes)):
    m = min(len(codes), n - done)
    if m < len(codes):
        b = synth.pack_batch(codes[:m])
    hs.build_packed(done, m, b["packed"], b["n_bases"], b["seg_seq"], b["seg_start"], b["seg_end"], b["eff_len"], b["one_mers"])
FAST = sum(1 << b for b in (2, 3, 5, 9, 13, 18, 21, 27, 28))
for name, mask in (("fast (9)", FAST), ("slow (11)", FAST | (1 << 7) | (1 << 29)), ("jensen_shannon only", 1 << 29)):
    ts = []
    for it in range(4):
        t0 = time.perf_counter()
        api.pair_features_raw(ctx, hs, None, hs, 3 + it, mask, m=n)
        ts.append((time.perf_counter() - t0, ctx.last_kernel_ms()[0]))
    wall = np.median([a for a, _ in ts[1:]]); tiles = np.median([b_ for _, b_ in ts[1:]])
    print(json.dumps({"features": name, "m": n, "wall_ms": round(wall * 1e3, 2), "tiles_ms": round(float(tiles), 3), "pairs_per_s_kernel": round(n / tiles * 1e3),
                      "cand_GBps": round(n * 4 ** k * dt / 8 / tiles / 1e6, 1)}), flush=True)
