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


DELTA = 3


def _centres():
    """14 centres, members of the same family next to each other so that the merge scan finds partners within DELTA"""
    seqs, headers = synth.families(77, N, 1000, family=6)
    order = sorted(range(N), key=lambda i: len(seqs[i]))          # position in _seqs() -> original index
    fam = [headers[i].split()[-1] for i in order]
    pos = sorted(range(N), key=lambda p: (fam[p], p))
    return pos[:14]


CENTRES = _centres()


def _centre_round(dist, rank, world, seqs, pred, nb):
    """update-round exchange: every rank contributes the centres it owns, all-gathers, and scans Trainer::merge locally"""
    import ctypes as C
    import torch
    from oracle import oracle_py
    plan = shard.ShardPlan(len(CENTRES), world, block=2)
    own = [oracle_py.hist(seqs[CENTRES[j]], K, DT) for j in plan.local_globals(rank)]

    class Backend:
        def centre_payload(self, n_pad):
            bins = torch.zeros(n_pad, nb, dtype=torch.int32)
            meta = torch.zeros(n_pad, 2, dtype=torch.int64)
            for i, h in enumerate(own):
                bins[i] = torch.from_numpy(h.array().astype(np.int32))
                meta[i] = torch.tensor([h.mag, h.length])
            return [bins, meta]

        def gather_buffers(self, n_rows):
            self.bins, self.meta = torch.full((n_rows, nb), -1, dtype=torch.int32), torch.zeros(n_rows, 2, dtype=torch.int64)
            return [self.bins, self.meta]

        def import_centres(self, rows):
            self.keep, self.centres = [], []
            for r in rows:
                a = np.ascontiguousarray(self.bins[r].numpy().astype(np.uint16))
                h = oracle_py.Hist()
                h.dtype, h.k, h.nbins = DT, K, nb
                h.bins = a.ctypes.data_as(C.c_void_p).value
                h.mag, h.length = int(self.meta[r, 0]), int(self.meta[r, 1])
                self.keep.append(a)
                self.centres.append(h)

    be = Backend()
    rows = shard.ShardedCentres(dist, plan, be, rank).gather()
    n = len(CENTRES)
    assert len(set(rows.tolist())) == n
    return [oracle_py.merge(pred, CUTOFF, be.centres, i, i + 1, min(n - 1, i + DELTA)) for i in range(n - 1)]


BLOCK_Q, BLOCK_STEPS = 3, 4


def _block_queries(step, total):
    return [((step * BLOCK_Q + j) * 7) % total for j in range(BLOCK_Q)]


def _block_steps(dist, rank, plan, hists, pred, nb):
    """the all-pairs exchange, double-buffered as bench.py drives it: block s + 1 is issued before block s is scored"""
    import ctypes as C
    import torch
    from oracle import oracle_py

    class Backend:
        def __init__(self):
            self.bins = [torch.zeros(nb, dtype=torch.int32) for _ in range(2 * BLOCK_Q)]
            self.meta = [torch.zeros(2, dtype=torch.int64) for _ in range(2 * BLOCK_Q)]
            self.q = [None] * (2 * BLOCK_Q)
            self.keep = [None] * (2 * BLOCK_Q)

        def query_buffers(self, j):
            return [self.bins[j], self.meta[j]]

        def export_query(self, local):
            h = hists[local]
            return [torch.from_numpy(h.array().astype(np.int32)), torch.tensor([h.mag, h.length], dtype=torch.int64)]

        def import_queries(self, n, base=0):
            for j in range(base, base + n):
                h = oracle_py.Hist()
                self.keep[j] = self.bins[j].numpy().astype(np.uint16)
                h.dtype, h.k, h.nbins = DT, K, nb
                h.bins = self.keep[j].ctypes.data_as(C.c_void_p).value
                h.mag, h.length = int(self.meta[j][0]), int(self.meta[j][1])
                self.q[j] = h

        def score_block(self, n, base=0):
            return np.stack([oracle_py.get_close(pred, CUTOFF, self.q[j], hists)[0] for j in range(base, base + n)])

    blk = shard.ShardedBlockScorer(dist, plan, Backend(), rank)
    pending = [None, None]
    totals = []
    for st in range(BLOCK_STEPS):
        cur, nxt = st % 2, (st + 1) % 2
        if pending[cur] is None:
            pending[cur] = blk.begin(_block_queries(st, plan.n_total), base=cur * BLOCK_Q)
        blk.finish(pending[cur], BLOCK_Q, base=cur * BLOCK_Q)
        pending[cur] = None
        pending[nxt] = blk.begin(_block_queries(st + 1, plan.n_total), base=nxt * BLOCK_Q)
        close, total = blk.score(BLOCK_Q, base=cur * BLOCK_Q)
        assert close.shape == (BLOCK_Q, len(hists))
        totals.append(total.tolist())
    for half in pending:
        for w in half or []:
            w.wait()
    # the one-shot form gives the same answer
    _, again = blk.score_block(_block_queries(1, plan.n_total))
    assert again.tolist() == totals[1]
    return totals


PACKED_PER_RANK, PACKED_STEPS = 2, 3


def _packed_first(step, m_min):
    return (step * PACKED_PER_RANK * 5) % (m_min - PACKED_PER_RANK + 1)


def _packed_steps(dist, rank, plan, hists, pred, nb):
    """the block exchange as bench.py drives it since r02: every rank contributes PACKED_PER_RANK consecutive local histograms
    and ONE all-gather per payload region (2 collectives per step) assembles the block in contiguous query buffers;
    double-buffered, block s + 1 issued before block s is scored"""
    import ctypes as C
    import torch
    from oracle import oracle_py
    nq = PACKED_PER_RANK * plan.world
    m_min = min(plan.local_count(r) for r in range(plan.world))
    calls = {"all_gather_into_tensor": 0, "broadcast": 0}

    class CountingDist:
        def __getattr__(self, name):
            fn = getattr(dist, name)
            if name in calls:
                def counted(*a, **kw):
                    calls[name] += 1
                    return fn(*a, **kw)
                return counted
            return fn

    own_bins = torch.stack([torch.from_numpy(h.array().astype(np.int32)) for h in hists])
    own_meta = torch.tensor([[h.mag, h.length] for h in hists], dtype=torch.int64)

    class Backend:
        def __init__(self):
            self.bins = torch.zeros(2 * nq, nb, dtype=torch.int32)
            self.meta = torch.zeros(2 * nq, 2, dtype=torch.int64)
            self.q = [None] * (2 * nq)
            self.keep = [None] * (2 * nq)

        def export_block(self, local_first, n):
            return [own_bins[local_first:local_first + n], own_meta[local_first:local_first + n]]

        def block_buffers(self, base, n_rows):
            return [self.bins[base:base + n_rows], self.meta[base:base + n_rows]]

        def import_queries(self, n, base=0):
            for j in range(base, base + n):
                h = oracle_py.Hist()
                self.keep[j] = self.bins[j].numpy().astype(np.uint16)
                h.dtype, h.k, h.nbins = DT, K, nb
                h.bins = self.keep[j].ctypes.data_as(C.c_void_p).value
                h.mag, h.length = int(self.meta[j][0]), int(self.meta[j][1])
                self.q[j] = h

        def score_block(self, n, base=0):
            return np.stack([oracle_py.get_close(pred, CUTOFF, self.q[j], hists)[0] for j in range(base, base + n)])

    blk = shard.ShardedBlockScorer(CountingDist(), plan, Backend(), rank)
    pending = [None, None]
    out = []
    for st in range(PACKED_STEPS):
        cur, nxt = st % 2, (st + 1) % 2
        if pending[cur] is None:
            pending[cur] = blk.begin_packed(_packed_first(st, m_min), PACKED_PER_RANK, base=cur * nq)
        blk.finish(pending[cur], nq, base=cur * nq)
        pending[cur] = None
        pending[nxt] = blk.begin_packed(_packed_first(st + 1, m_min), PACKED_PER_RANK, base=nxt * nq)
        # odd steps leave the counts' all-gather in flight (bench.py's way: nothing in the step needs them), even steps wait for it
        close, total = blk.score(nq, base=cur * nq, defer=bool(st % 2))
        assert close.shape == (nq, len(hists))
        out.append((blk.packed_globals(_packed_first(st, m_min), PACKED_PER_RANK), total))
    for half in pending:
        for w in half or []:
            w.wait()
    out = [(g, (t.total() if hasattr(t, "total") else t).tolist()) for g, t in out]
    # one all-gather per payload region and block (PACKED_STEPS + 1 blocks were issued) + one of the counts per scored block, no
    # per-query broadcast
    assert calls["all_gather_into_tensor"] == 2 * (PACKED_STEPS + 1) + PACKED_STEPS and calls["broadcast"] == 0, calls
    return out


ROW_Q, ROW_STEPS = 6, 3


def _row_queries(step, total):
    return [((step * ROW_Q + j) * 11) % total for j in range(ROW_Q)]


def _row_steps(dist, rank, plan, hists, pred, nb):
    """the all-pairs job sharded by QUERY rows (bench.py's default since r05): one set-up all-gather replicates every rank's
    histograms' source rows, each rank rebuilds ALL candidates locally, then scores its share of each step's rows against all
    of them; the only per-step collective is the counts' all-gather (left in flight on odd steps)"""
    import ctypes as C
    import torch
    from oracle import oracle_py
    calls = {"all_gather_into_tensor": 0, "broadcast": 0, "all_gather": 0}

    class CountingDist:
        def __getattr__(self, name):
            fn = getattr(dist, name)
            if name in calls:
                def counted(*a, **kw):
                    calls[name] += 1
                    return fn(*a, **kw)
                return counted
            return fn

    class Backend:
        def shard_payload(self, n_pad):
            bins = torch.zeros(n_pad, nb, dtype=torch.int32)
            meta = torch.zeros(n_pad, 2, dtype=torch.int64)
            for i, h in enumerate(hists):
                bins[i] = torch.from_numpy(h.array().astype(np.int32))
                meta[i] = torch.tensor([h.mag, h.length])
            return [bins, meta]

        def gather_buffers(self, n_rows):
            self.bins, self.meta = torch.full((n_rows, nb), -1, dtype=torch.int32), torch.zeros(n_rows, 2, dtype=torch.int64)
            return [self.bins, self.meta]

        def import_rows(self, rows):
            self.keep, self.all = [], []
            for r in rows:
                a = np.ascontiguousarray(self.bins[r].numpy().astype(np.uint16))
                h = oracle_py.Hist()
                h.dtype, h.k, h.nbins = DT, K, nb
                h.bins = a.ctypes.data_as(C.c_void_p).value
                h.mag, h.length = int(self.meta[r, 0]), int(self.meta[r, 1])
                self.keep.append(a)
                self.all.append(h)

        def score_rows(self, globals_):
            return [int(oracle_py.get_close(pred, CUTOFF, self.all[g], self.all)[0].sum()) for g in globals_]

    be = Backend()
    rr = shard.ReplicatedRows(CountingDist(), plan, be, rank)
    rows = rr.replicate()
    assert len(set(rows.tolist())) == plan.n_total and len(be.all) == plan.n_total
    setup = dict(calls)
    out = []
    for st in range(ROW_STEPS):
        qs = _row_queries(st, plan.n_total)
        assert sum(len(rr.rows_of(qs, r)) for r in range(plan.world)) == ROW_Q
        out.append(rr.score(qs, defer=bool(st % 2)))
    out = [(t.total() if hasattr(t, "total") else t).tolist() for t in out]
    # set-up: one all-gather per payload region; per step: the counts only
    assert setup["all_gather_into_tensor"] == 2 and calls["all_gather_into_tensor"] == 2 + ROW_STEPS and calls["broadcast"] == 0 and calls["all_gather"] == 0, calls
    return out


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
        q.put((rank, mine.tolist(), out, _centre_round(dist, rank, world, seqs, pred, nb), _block_steps(dist, rank, plan, hists, pred, nb),
               _packed_steps(dist, rank, plan, hists, pred, nb), _row_steps(dist, rank, plan, hists, pred, nb)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_passes_match_single_process(oracle, world):
    import torch.multiprocessing as mp
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
        r, mine, out, merged, blocks, packed, by_rows = q.get(timeout=180)
        res[r] = (mine, out, merged, blocks, packed, by_rows)
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
            mine, out = res[r][0], res[r][1]
            _, flags, g, sim, is_min, n_close = out[qi]
            glob_flags[np.array(mine)] = flags
            assert (g, is_min, n_close) == (bp, im, int(f.sum())) and sim == pytest.approx(bs, rel=1e-12)
        assert np.array_equal(glob_flags, f)
    # every rank holds all centres after the all-gather: its merge scan equals the single-process one
    cs = [hs[c] for c in CENTRES]
    want = [oracle.merge(pred, CUTOFF, cs, i, i + 1, min(len(cs) - 1, i + DELTA)) for i in range(len(cs) - 1)]
    assert any(w > i for i, w in enumerate(want))           # the scan does merge something
    for r in range(world):
        assert res[r][2] == want
    # double-buffered all-pairs blocks: per-query close counts over every rank == the single-process count
    for st in range(BLOCK_STEPS):
        want_counts = [float(oracle.get_close(pred, CUTOFF, hs[g], hs)[0].sum()) for g in _block_queries(st, len(seqs))]
        for r in range(world):
            assert res[r][3][st] == want_counts, (st, r)
    # packed exchange (2 collectives per block): every rank reports the same block composition and the single-process counts
    for st in range(PACKED_STEPS):
        globals_, _ = res[0][4][st]
        assert len(set(globals_)) == PACKED_PER_RANK * world
        want_counts = [float(oracle.get_close(pred, CUTOFF, hs[g], hs)[0].sum()) for g in globals_]
        for r in range(world):
            assert res[r][4][st] == (globals_, want_counts), (st, r)
    # sharded by query rows over replicated candidates: every rank reports the single-process counts of the step's whole query list
    for st in range(ROW_STEPS):
        want_counts = [int(oracle.get_close(pred, CUTOFF, hs[g], hs)[0].sum()) for g in _row_queries(st, len(seqs))]
        for r in range(world):
            assert res[r][5][st] == want_counts, (st, r)
