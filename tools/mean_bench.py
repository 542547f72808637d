#!/usr/bin/env python3
"""get_mean / closest (k_colsum + the members streamed against the rounded mean + k_distance_d) on m members -- run on the GPU box.
   python tools/mean_bench.py [m] [k] [dtype] [reps]
Algorithmic bytes per call: 2 * m * 4^k * sizeof(T): one pass for the mean, one for distance_d."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from meshclust2_amd import api, synth
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 9
dtype = int(sys.argv[3]) if len(sys.argv) > 3 else 32
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
ctx = api.Context(0)
codes, _ = synth.family_codes(99, 4 * m, 1000)
hs = api.HistogramSet(ctx, k, dtype, 4 * m)
b = synth.pack_batch(codes)
hs.build_packed(0, 4 * m, b["packed"], b["n_bases"], b["seg_seq"], b["seg_start"], b["seg_end"], b["eff_len"], b["one_mers"])
rng = np.random.default_rng(1)
walls = []
for r in range(reps + 2):
    mem = rng.permutation(4 * m)[:m].astype(np.uint32)
    ctx.synchronize()
    t0 = time.perf_counter()
    api.mean_nearest(ctx, hs, mem)
    walls.append(time.perf_counter() - t0)
w = float(np.median(walls[2:]))
alg = 2 * m * (4 ** k) * dtype // 8
print(json.dumps({"case": "mean_nearest", "m": m, "k": k, "dtype": dtype, "wall_ms": round(w * 1e3, 3), "algorithmic_bytes_per_call": alg,
                  "alg_GBps_wall": round(alg / w / 1e9, 1),
                  "roofline": {"profile_key": "mean_nearest,m=%d,k=%d,dtype=%d" % (m, k, dtype)}}), flush=True)
