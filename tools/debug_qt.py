import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch  # noqa
from meshclust2_amd import api, synth
from golden_util import FEATS, weights_text
from test_gpu_qxm_direct import _repeat_bearing, FAST_MASK
ctx = api.Context(0)
dtype, k, n, length, nq, repeats = 32, 5, 60, 100, 33, "unit3"
seqs, _ = synth.families(4100 + 31 * k + dtype + nq, n, length, family=6, length_jitter=length // 10)
seqs = _repeat_bearing(seqs, 7, repeats)
hs = api.HistogramSet(ctx, k, dtype, n)
hs.build(seqs)
rng = np.random.default_rng(k * 1000 + nq)
q_slots = rng.integers(0, n, nq).astype(np.uint32)
q_slots[:3] = (1, 8, 0)
cands = np.arange(n, dtype=np.uint32)
got = api.score_multi(ctx, None, hs, cands, hs, q_slots, feat_mask=FAST_MASK, want=())
print(ctx.last_kernel_info())
H = np.stack([hs.download(i).astype(np.int64) for i in range(n)])
bad = 0
for qi, q in enumerate(q_slots):
    for c in range(n):
        dot = int((H[q] * H[c]).sum())
        eu = np.sqrt(float(((H[q] - H[c]) ** 2).sum()))
        if abs(got["raw"][qi][c][1] - eu) > 1e-9:
            bad += 1
            if bad < 6:
                both = np.flatnonzero((H[q] >= 3) & (H[c] >= 2))
                print("bad", qi, int(q), c, got["raw"][qi][c][1], eu, "bins large in q & present in c:", both[:10], H[q][both[:10]], H[c][both[:10]])
print("bad pairs", bad, "of", nq * n)
