#!/usr/bin/env python3
"""RCCL collectives straight on the library's device memory (run on the GPU box, one rank: the pool's boxes have one GPU).
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_views_check.py
Checks that the nccl (= RCCL) backend accepts the torch views shard.device_tensors() makes over hipMalloc'ed histogram slots for
broadcast / all_gather / all_gather_into_tensor, and that the slots score the same afterwards. The N > 1 logic itself is covered
by tests/test_shard_gloo.py on CPU."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from meshclust2_amd import api, shard, synth

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
ctx = api.Context(0)
seqs, _ = synth.families(99, 40, 1000)
hs = api.HistogramSet(ctx, 9, 32, 64)
hs.build(seqs)
feat = api.Feature.from_text(ctx, open(os.path.join(ROOT, "tests", "golden", "weights_k9_u32.txt")).read(), 0)
q = np.arange(16, dtype=np.uint32)
want = api.score_multi(ctx, feat, hs, np.arange(40, dtype=np.uint32), hs, q, want=("sum",))["sum"]
ctx.synchronize()
bins, scal = shard.device_tensors(hs, 64)
pending = []
for j in range(8):                                    # "query broadcast" into slots 40..47 from slots 0..7, asynchronous as bench.py issues them
    for t in (bins, scal):
        t[40 + j].copy_(t[j])
        pending.append(dist.broadcast(t[40 + j], src=0, async_op=True))
busy = api.score_multi(ctx, feat, hs, np.arange(40, dtype=np.uint32), hs, q, want=("sum",))["sum"]      # the library works underneath the collectives
assert np.array_equal(busy, want)
for w in pending:
    w.wait()
for t in (bins, scal):                                # "centre all-gather" of slots 8..15 into 48..55
    dist.all_gather_into_tensor(t[48:56], t[8:16])
rec = torch.tensor([1.0, 0.5, 7.0], dtype=torch.float64, device="cuda")
out = [torch.zeros_like(rec)]
dist.all_gather(out, rec)
torch.cuda.synchronize()
hs.import_done(40, 16)
got = api.score_multi(ctx, feat, hs, np.arange(40, dtype=np.uint32), hs, np.arange(40, 56, dtype=np.uint32), want=("sum",))["sum"]
assert np.array_equal(got, want), "slots moved by RCCL score differently"
assert out[0].tolist() == [1.0, 0.5, 7.0]
dist.barrier()
dist.destroy_process_group()
print("rccl views ok")
