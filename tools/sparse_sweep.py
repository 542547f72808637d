#!/usr/bin/env python3
"""pairs/s of the sparse layout vs the dense one (1 x M get_close passes) -- run on the GPU box.
  python tools/sparse_sweep.py [n] [k] [dtype] [length]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshclust2_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 9
dt = int(sys.argv[3]) if len(sys.argv) > 3 else 32
length = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
ctx = api.Context(0)
base = min(n, 4000)
codes = [synth.member(5, t // 20, t % 20, synth.template(5, t // 20, length)) for t in range(base)]
b = synth.pack_batch(codes)
t0 = time.perf_counter()
hs = api.HistogramSet(ctx, k, dt, n, sparse_entries=int(n * (length + 20)))
done = 0
while done < n:
    m = min(base, n - done)
    if m < base:
        b = synth.pack_batch(codes[:m])
    hs.build_packed(done, m, b["packed"], b["n_bases"], b["seg_seq"], b["seg_start"], b["seg_end"], b["eff_len"], b["one_mers"])
    done += m
build_s = time.perf_counter() - t0
wts = "weights_k9_u32.txt" if dt == 32 else "weights_k5_u16.txt"
feat = api.Feature.from_text(ctx, open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", wts)).read(), 0)
trn = api.Trainer(ctx, feat, 0.5)
ms = []
for it in range(6):
    t0 = time.perf_counter()
    trn.get_close(hs, None, hs, it, m=n)
    ms.append((ctx.last_kernel_ms()[0], (time.perf_counter() - t0) * 1e3))
kern = float(np.median([a for a, _ in ms[1:]])); wall = float(np.median([w for _, w in ms[1:]]))
ent = sum(hs.entries(i) for i in range(min(n, 200))) / min(n, 200)
print(json.dumps({"layout": "sparse", "n": n, "k": k, "dtype": dt, "length": length, "entries_per_slot": round(ent), "build_s": round(build_s, 2),
                  "bytes_resident": hs.nbytes(), "kernel_ms": round(kern, 3), "wall_ms": round(wall, 3), "pairs_per_s_kernel": round(n / kern * 1e3),
                  "pairs_per_s_wall": round(n / wall * 1e3), "list_GBps": round(n * ent * 12 / kern / 1e6, 1)}), flush=True)
