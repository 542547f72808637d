#!/usr/bin/env python3
"""Cross-check of the sparse pair kernels on a mixed-length set -- run on the GPU box, twice:
   python tools/sparse_crosscheck.py > a.txt;  MSC_SPARSE_NO_MP=1 python tools/sparse_crosscheck.py > b.txt;  cmp a.txt b.txt
One line per (query, candidate): the 11 raw statistics (floats to 10 significant digits: the divergence sums are folded in a
different order by the two kernels, so a last-digit difference in those two columns on a line or two out of 10^4 is expected; r01:
identical at k=9/u8 and k=13/u64, one such line at k=11/u16). The merge-path kernel (k_pair_sparse_mp) against the lane-per-sub-range kernel
(k_pair_sparse, 64-bit running values) over thousands of list pairs of every length ratio."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshclust2_amd import api
k = int(sys.argv[1]) if len(sys.argv) > 1 else 11
dtype = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
rng = np.random.default_rng(7)
alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
base = rng.choice(alpha, 40000).tobytes()
seqs = []
for i in range(n):
    L = int(np.exp(rng.uniform(np.log(30), np.log(30000))))
    if i % 3 == 0:                                   # relatives of one template: shared bins, ties on the merge path
        a = int(rng.integers(0, 40000 - L))
        s = bytearray(base[a:a + L])
        for _ in range(L // 40):
            s[int(rng.integers(0, L))] = int(alpha[int(rng.integers(0, 4))])
    else:
        s = bytearray(rng.choice(alpha, L, p=[0.45, 0.05, 0.05, 0.45] if i % 7 == 0 else None).tobytes())
    seqs.append(bytes(s))
seqs[5] = b"ACG"                                     # no k-mers: an empty list
ctx = api.Context(0)
hs = api.HistogramSet(ctx, k, dtype, n, sparse_entries=sum(len(s) for s in seqs) + 4096)
hs.build(seqs)
mask = 0
for b in (2, 3, 5, 7, 9, 13, 18, 21, 27, 28, 29):
    mask |= 1 << b
for q in (0, 1, 5, 6, 7, n - 1, n // 2):
    raw = api.pair_features_raw(ctx, hs, None, hs, q, mask, api.ORDER_CAND_FIRST, m=n)
    for c in range(n):
        print(q, c, " ".join("%.10g" % v for v in raw[c]))
