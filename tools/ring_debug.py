import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from meshclust2_amd import api, synth
from golden_util import FEAT_BIT
golden = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
ctx = api.Context(0)
dtype, k, n, length = 32, 9, 330, 1000
seqs, _ = synth.families(900 + k, n, length, family=10)
hs = api.HistogramSet(ctx, k, dtype, len(seqs)); hs.build(seqs)
feat = api.Feature.from_text(ctx, open(os.path.join(golden, "weights_k9_u32.txt")).read(), 0)
mask = (1 << FEAT_BIT["manhattan"]) | (1 << FEAT_BIT["emd"]) | (1 << FEAT_BIT["normalized_vectors"])
for nq in (8, 9, 16):
    cands = np.arange(5, n, dtype=np.uint32)
    qs = (np.arange(nq, dtype=np.uint32) * 3) % n
    multi = api.score_multi(ctx, feat, hs, cands, hs, qs, feat_mask=mask)
    for i, q in enumerate(qs):
        raw = api.pair_features_raw(ctx, hs, cands, hs, int(q), mask)
        bad = np.argwhere(multi["raw"][i] != raw)
        print(nq, i, "bad", len(bad), bad[:6].tolist(), flush=True)
        if len(bad):
            c, f = bad[0]
            print("   got", multi["raw"][i][c], "want", raw[c])
