#!/usr/bin/env python3
"""Latency of ONE Trainer::get_close call (the unit of the step-serial accumulate loop, cluster/ClusterFactory.cpp:553-610) by
window size -- run on the GPU box.   python tools/step_time.py [k] [dtype] [n_points] [length] [dense|sparse]
A step = slot list up, streaming kernel + epilogue + reduce back to back on the stream, (record, flags) down, one stream sync.
Prints wall microseconds per call (median of 200) and the streaming kernel's share; windows are random subsets of the set."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from meshclust2_amd import api, synth
k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dtype = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
length = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
layout = sys.argv[5] if len(sys.argv) > 5 else "dense"
ctx = api.Context(0)
codes, _ = synth.family_codes(31, n, length)
hs = api.HistogramSet(ctx, k, dtype, n, sparse_entries=int(n * (length + 64)) if layout == "sparse" else 0)
b = synth.pack_batch(codes)
hs.build_packed(0, n, b["packed"], b["n_bases"], b["seg_seq"], b["seg_start"], b["seg_end"], b["eff_len"], b["one_mers"])
wts = "weights_k8_u16.txt" if k == 8 else "weights_k9_u32.txt" if k == 9 else "weights_k5_u16.txt"
feat = api.Feature.from_text(ctx, open(os.path.join(ROOT, "tests", "golden", wts)).read(), 0)
trn = api.Trainer(ctx, feat, 0.5)
rng = np.random.default_rng(3)
out = []
for m in sorted(set(min(x, n) for x in (0, 1, 16, 128, 1024, 8192, n))):
    w = rng.permutation(n)[:m].astype(np.uint32)
    row = {"window": m, "layout": layout, "length": length, "window_bytes": m * (4 ** k) * dtype // 8}
    for timing in (True, False):          # with / without the event records behind msc_last_kernel_ms (msc_set_kernel_timing)
        ctx.set_kernel_timing(timing)
        walls, kern = [], []
        for rep in range(220):
            t0 = time.perf_counter()
            trn.get_close(hs, w, hs, rep % n)
            walls.append(time.perf_counter() - t0)
            if m and timing:
                kern.append(ctx.last_kernel_ms()[0])
        row["wall_us" if timing else "wall_us_no_events"] = round(float(np.median(walls[20:])) * 1e6, 1)
        if timing:
            row["kernel_us"] = round(float(np.median(kern[20:])) * 1e3, 1) if m else 0.0
    out.append(row)
print(json.dumps({"case": "get_close step latency", "k": k, "dtype": dtype, "points": n, "steps": out}), flush=True)
