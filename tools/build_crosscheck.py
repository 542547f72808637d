#!/usr/bin/env python3
"""Cross-check of the histogram builders on a mixed-length batch -- run on the GPU box, twice:
   python tools/build_crosscheck.py > a.txt;  MSC_NO_SORT_DENSE_BUILD=1 python tools/build_crosscheck.py > b.txt;  cmp a.txt b.txt
Prints one line per slot (hash of the downloaded bins + the scalar record), so the sort builder (k_build_sort) and the
fill + count + finalize path can be compared over thousands of sequences, not only the handful the oracle tests hold."""
import hashlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshclust2_amd import api
k = int(sys.argv[1]) if len(sys.argv) > 1 else 9
dtype = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
rng = np.random.default_rng(99)
alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
seqs = []
for i in range(n):
    L = int(np.exp(rng.uniform(np.log(200), np.log(30000))))
    s = bytearray(rng.choice(alpha, L, p=[0.4, 0.1, 0.1, 0.4] if i % 5 == 0 else None).tobytes())
    if i % 11 == 0:
        a = int(rng.integers(0, max(1, L - 300)))
        s[a:a + 260] = b"A" * min(260, L - a)          # a run that saturates uint8_t
    if i % 13 == 0:
        a = int(rng.integers(0, max(1, L - 50)))
        s[a:a + 25] = b"N" * min(25, L - a)
    seqs.append(bytes(s))
ctx = api.Context(0)
hs = api.HistogramSet(ctx, k, dtype, n)
hs.build(seqs)
emd = api.pair_features_raw(ctx, hs, None, hs, 1, 1 << 18, api.ORDER_CAND_FIRST, m=n)[:, 0]      # exercises the tile prefixes of every slot
for i in range(n):
    inf = hs.info(i)
    h = hashlib.sha1(hs.download(i).tobytes()).hexdigest()[:16]
    print(i, len(seqs[i]), h, inf["mag"], inf["length"], inf["overflow"], inf["one_mers"], inf.get("sum"), inf.get("sum_sq"), inf.get("max_count"), inf.get("stddev"), emd[i])
