"""The N > 1 path on CPU: world_size-2 gloo processes shard the candidates, broadcast the query and all-gather the
24-byte records through meshclust2_amd/shard.py; the per-rank scorer is the CPU oracle (test infrastructure).
The folded result must equal the single-process oracle answer for the whole window."""
import os
import socket
import sys

import numpy as np
import pytest

from golden_util import weights_text
from meshclust2_amd import shard, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K, DT, N, CUTOFF = 5, 16, 46, 0.9


def test_plan_is_a_bijection():
    for n, w, b in ((46, 2, 8), (1000, 8, 100), (7, 3, 1000), (2001, 4, 1000)):
        plan = shard.ShardPlan(n, w, b)
        seen = set()
        for g in range(n):
            r, l = plan.owner(g), plan.local(g)
            assert plan.global_index(r, l) == g and l < plan.local_count(r)
            seen.add((r, l))
        assert len(seen) == n and sum(plan.local_count(r) for r in range(w)) == n


def test_fold_follows_serial_tie_rule():
    assert shard.fold_records([(0, -1.0, -1), (0, -1.0, -1)]) == (0, -1.0, -1, True)
    assert shard.fold_records([(2, 0.7, 12), (1, 0.7, 5)]) == (3, 0.7, 5, False)          # equal maxima: first in window order
    assert shard.fold_records([(0, 0.2, 3), (0, 0.9, 40)]) == (0, 0.9, 40, True)


def _seqs():
    seqs, _ = synth.families(77, N, 1000, family=6)
    seqs = sorted(seqs, key=len)
    return seqs


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle import oracle_py
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seqs = _seqs()
        plan = shard.ShardPlan(len(seqs), world, block=8)
        mine = plan.local_globals(rank)
        hists = [oracle_py.hist(seqs[g], K, DT) for g in mine]
        pred = oracle_py.predictor(weights_text("weights_k5_u16.txt"))
        nb = 4 ** K

        class Backend:
            """CPU stand-in for the GPU library: payload = bins + (mag, length) like the device scalar record"""
            def __init__(self):
                self.bins = torch.zeros(nb, dtype=torch.int32)
                self.meta = torch.zeros(2, dtype=torch.int64)
                self.query = None

            def query_buffers(self):
                return [self.bins, self.meta]

            def export_query(self, local):
                h = hists[local]
                return [torch.from_numpy(h.array().astype(np.int32)), torch.tensor([h.mag, h.length], dtype=torch.int64)]

            def import_query(self):
                import ctypes as C
                h = oracle_py.Hist()
                self._keep = self.bins.numpy().astype(np.uint16)
                h.dtype, h.k, h.nbins = DT, K, nb
                h.bins = self._keep.ctypes.data_as(C.c_void_p).value
                h.mag, h.length = int(self.meta[0]), int(self.meta[1])
                self.query = h

            def score_local(self):
                f, bp, bs, _ = oracle_py.get_close(pred, CUTOFF, self.query, hists)
                return f, bp, bs

        trn = shard.ShardedTrainer(dist, plan, Backend(), rank)
        out = []
        for qg in (0, 9, 17, 30, 45):
            flags, g, sim, is_min, n_close = trn.get_close(qg)
            out.append((qg, flags.tolist(), g, sim, is_min, n_close))
        q.put((rank, mine.tolist(), out))
    finally:
        dist.destroy_process_group()


def test_world2_matches_single_process(oracle):
    import torch.multiprocessing as mp
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, mine, out = q.get(timeout=180)
        res[r] = (mine, out)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    seqs = _seqs()
    hs = [oracle.hist(s_, K, DT) for s_ in seqs]
    pred = oracle.predictor(weights_text("weights_k5_u16.txt"))
    for qi, qg in enumerate((0, 9, 17, 30, 45)):
        # the query scores against itself too on its owner, like a centre that is still in the window
        f, bp, bs, im = oracle.get_close(pred, CUTOFF, hs[qg], hs)
        glob_flags = np.zeros(len(seqs), dtype=np.uint8)
        for r in range(world):
            mine, out = res[r]
            _, flags, g, sim, is_min, n_close = out[qi]
            glob_flags[np.array(mine)] = flags
            assert (g, is_min, n_close) == (bp, im, int(f.sum())) and sim == pytest.approx(bs, rel=1e-12)
        assert np.array_equal(glob_flags, f)
