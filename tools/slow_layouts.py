#!/usr/bin/env python3
"""A --feat slow model (jefferey + jensen_shannon) through one get_close pass over 100 000 x 1 kb histograms (k=9, uint16_t) in the dense
and in the sparse layout -- run on the GPU box.   python tools/slow_layouts.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from meshclust2_amd import api, synth
n, k, dt, L = 100000, 9, 16, 1000
ctx = api.Context(0)
codes = [synth.member(5, t // 20, t % 20, synth.template(5, t // 20, L)) for t in range(4000)]
b = synth.pack_batch(codes)
feat = api.Feature.from_text(ctx, open(os.path.join(ROOT, "tests", "golden", "weights_k5_u16_slow.txt")).read(), 0)
trn = api.Trainer(ctx, feat, 0.5)
for layout in ("dense", "sparse"):
    hs = api.HistogramSet(ctx, k, dt, n, sparse_entries=int(n * (L + 20)) if layout == "sparse" else 0)
    for d in range(0, n, 4000):
        hs.build_packed(d, 4000, b["packed"], b["n_bases"], b["seg_seq"], b["seg_start"], b["seg_end"], b["eff_len"], b["one_mers"])
    ms = []
    for it in range(5):
        t0 = time.perf_counter()
        trn.get_close(hs, None, hs, it, m=n)
        ms.append(((time.perf_counter() - t0) * 1e3, ctx.last_kernel_ms()[0]))
    w = np.median([a for a, _ in ms[1:]]); kk = np.median([c for _, c in ms[1:]])
    print("%s slow model (jefferey + jensen_shannon): get_close over %d candidates: %.2f ms wall (%.1f M pairs/s), streaming kernel %.2f ms" % (layout, n, w, n / w / 1e3, kk), flush=True)
    hs.close()
