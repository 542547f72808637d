#!/usr/bin/env python3
"""Device time of msc_hist_build_packed on synthetic sequences -- run on the GPU box.
   python tools/build_time.py [n_seqs] [length] [k] [dtype]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from meshclust2_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
length = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 9
dtype = int(sys.argv[4]) if len(sys.argv) > 4 else 32
ctx = api.Context(0)
seqs, _ = synth.families(4242, n, length)
hs = api.HistogramSet(ctx, k, dtype, n)
for rep in range(3):
    ctx.synchronize()
    t0 = time.perf_counter()
    hs.build(seqs)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print("build %d x %d bp k=%d u%d: %.1f ms  %.2f M seq/s  %.2f TB/s written (host encode + PCIe included)" %
          (n, length, k, dtype, dt * 1e3, n / dt / 1e6, n * (4 ** k) * dtype / 8 / dt / 1e12), flush=True)
