#!/usr/bin/env python3
"""Achieved algorithmic GB/s of the pair kernel (and of hist_build) over (k, dtype) -- run on the GPU box.
  python tools/kernel_sweep.py [--gib 8]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshclust2_amd import api, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gib", type=float, default=8.0)
ap.add_argument("--cases", default="9:32,9:16,9:8,7:32,7:16,5:16,5:32,11:8,4:8,9:64")
args = ap.parse_args()
ctx = api.Context(0)
for case in args.cases.split(","):
    k, dt = map(int, case.split(":"))
    hb = 4 ** k * dt // 8
    n = int(min(args.gib * 2 ** 30 / max(hb, 1024), 2_000_000))
    codes = [synth.member(5, t // 20, t % 20, synth.template(5, t // 20, 1000)) for t in range(min(n, 4000))]
    reps = (n + len(codes) - 1) // len(codes)
    hs = api.HistogramSet(ctx, k, dt, n)
    b = synth.pack_batch(codes)
    t0 = time.perf_counter()
    done = 0
    while done < n:
        m = min(len(codes), n - done)
        if m < len(codes):
            b = synth.pack_batch(codes[:m])
        hs.build_packed(done, m, b["packed"], b["n_bases"], b["seg_seq"], b["seg_start"], b["seg_end"], b["eff_len"], b["one_mers"])
        done += m
    build_s = time.perf_counter() - t0
    text = open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "weights_k5_u16.txt")).read()
    feat = api.Feature.from_text(ctx, text, 0)
    trn = api.Trainer(ctx, feat, 0.5)
    ms = []
    for it in range(6):
        t0 = time.perf_counter()
        trn.get_close(hs, None, hs, it, m=n)
        wall = time.perf_counter() - t0
        ms.append((ctx.last_kernel_ms()[0], wall * 1e3))
    tiles = float(np.median([a for a, _ in ms[1:]]))
    wall = float(np.median([b_ for _, b_ in ms[1:]]))
    print(json.dumps({"k": k, "dtype": dt, "n": n, "hist_bytes": hb, "pair_tiles_ms": round(tiles, 4), "get_close_wall_ms": round(wall, 3),
                      "alg_GBps": round(n * hb / tiles / 1e6, 1), "pairs_per_s_kernel": round(n / tiles * 1e3), "pairs_per_s_wall": round(n / wall * 1e3),
                      "build_seq_per_s_incl_host": round(n / build_s)}), flush=True)
    hs.close()
