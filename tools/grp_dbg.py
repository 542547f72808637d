import numpy as np, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from meshclust2_amd import api, synth
ctx = api.Context(0)
seqs, _ = synth.families(515, 12, 2000, family=4, length_jitter=100)
print("n", len(seqs), flush=True)
hs = api.HistogramSet(ctx, 8, 16, len(seqs)); hs.build(list(seqs))
ctx.synchronize()
print("built", flush=True)
for mask in ((1 << 2), (1 << 14)):
    r = api.pair_features_raw(ctx, hs, np.arange(len(seqs), dtype=np.uint32), hs, 3, mask)
    print("mask", hex(mask), np.asarray(r)[:2].tolist(), flush=True)
