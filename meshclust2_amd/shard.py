"""Sharding of the 1 x M scoring pass over ranks (one process per GPU) -- SURVEY.md 8(e).

Pairs are independent, so candidates are dealt to ranks and a pass needs only two tiny exchanges:
  1. the query histogram (bins + scalar record) is BROADCAST from the rank that owns it,
  2. one 24-byte record (n_close, best_sim, best_global_index) per rank is ALL-GATHERED and folded with the
     reference's serial arg-max rule (strict '>' in window order: the smallest global index among equal maxima,
     cluster/Trainer.cpp:26-37,59 at OMP_NUM_THREADS=1).
Histograms sorted by length are dealt block-cyclically (block = one bvec bin of 1000, cluster/CRunner.cpp:585) so
every length window is balanced over the ranks.

The module is backend-agnostic: `backend` supplies export_query / import_query / score_local. bench.py plugs in the
GPU library (payload = device tensors, collectives over RCCL); tests/test_shard_gloo.py plugs in CPU tensors over gloo.
"""
import numpy as np


class ShardPlan:
    """global index (position in the length-sorted order) <-> (rank, local slot), block-cyclic."""

    def __init__(self, n_total, world, block=1000):
        self.n_total, self.world, self.block = int(n_total), int(world), int(block)

    def owner(self, g):
        return (g // self.block) % self.world

    def local(self, g):
        b = g // self.block
        return (b // self.world) * self.block + g % self.block

    def global_index(self, rank, local):
        b_local, off = divmod(local, self.block)
        return (b_local * self.world + rank) * self.block + off

    def local_count(self, rank):
        full, rem = divmod(self.n_total, self.block)
        n = (full // self.world) * self.block
        extra = full % self.world
        if rank < extra:
            n += self.block
        elif rank == extra:
            n += rem
        return n

    def local_globals(self, rank):
        return np.array([self.global_index(rank, i) for i in range(self.local_count(rank))], dtype=np.int64)


def fold_records(records):
    """records: iterable of (n_close, best_sim, best_global_index or -1) -> (n_close_total, best_sim, best_global, is_min)"""
    n_close = 0
    best_sim, best_g = -1.0, -1
    for n, sim, g in records:
        n_close += int(n)
        g = int(g)
        if g < 0:
            continue
        if sim > best_sim or (sim == best_sim and (best_g < 0 or g < best_g)):
            best_sim, best_g = float(sim), g
    return n_close, best_sim, best_g, n_close == 0


class ShardedTrainer:
    """Trainer::get_close over sharded candidates.

    backend.export_query(local_slot) -> list of torch tensors holding the query payload (owner only)
    backend.query_buffers()          -> list of torch tensors every rank receives the payload into (same shapes)
    backend.import_query()           -> called after the broadcast landed in query_buffers()
    backend.score_local()            -> (close_flags[np.uint8, local_count], best_local_pos or -1, best_sim)
    """

    def __init__(self, dist, plan, backend, rank, device="cpu"):
        self.dist, self.plan, self.backend, self.rank, self.device = dist, plan, backend, rank, device

    def get_close(self, query_global):
        import torch
        owner = self.plan.owner(query_global)
        bufs = self.backend.query_buffers()
        if self.rank == owner:
            for dst, src in zip(bufs, self.backend.export_query(self.plan.local(query_global))):
                dst.copy_(src)
        if self.plan.world > 1:
            for b in bufs:
                self.dist.broadcast(b, src=owner)
        self.backend.import_query()
        flags, best_local, best_sim = self.backend.score_local()
        best_g = self.plan.global_index(self.rank, best_local) if best_local >= 0 else -1
        rec = torch.tensor([float(flags.sum()), float(best_sim), float(best_g)], dtype=torch.float64, device=self.device)
        if self.plan.world > 1:
            out = [torch.zeros(3, dtype=torch.float64, device=self.device) for _ in range(self.plan.world)]
            self.dist.all_gather(out, rec)
            recs = torch.stack(out).cpu().numpy()
        else:
            recs = rec.cpu().numpy().reshape(1, 3)
        n_close, sim, g, is_min = fold_records(recs)
        return flags, g, sim, is_min, n_close


class ShardedBlockScorer:
    """n_q queries x every candidate of every rank in one pass per rank (the all-pairs shape, fastcar work()).

    backend.query_buffers(j) / export_query(local) as above but per query buffer j; backend.import_queries(n, base);
    backend.score_block(n, base) -> close flags [n, local_count] (np.uint8) of the queries in buffers base .. base + n - 1.
    Per block the only exchanges are the query payload (begin_packed: one all-gather per payload region = 2 collectives per
    block; begin: one broadcast per query and region, for blocks whose queries are not spread evenly) and one all-gather of
    the per-query close counts.

    The exchange is split so that a caller can double-buffer it: begin() only ISSUES the copies and asynchronous broadcasts of a
    block into buffers base .., finish() waits for them and registers the slots, score() runs the local pass. bench.py issues
    block s + 1 before it scores block s, so the collectives run on RCCL's stream underneath the streaming kernel."""

    def __init__(self, dist, plan, backend, rank, device="cpu"):
        self.dist, self.plan, self.backend, self.rank, self.device = dist, plan, backend, rank, device

    def begin(self, query_globals, base=0):
        pending = []
        for j, qg in enumerate(query_globals):
            owner = self.plan.owner(qg)
            bufs = self.backend.query_buffers(base + j)
            if self.rank == owner:
                for dst, src in zip(bufs, self.backend.export_query(self.plan.local(qg))):
                    dst.copy_(src)
            if self.plan.world > 1:
                # all 2 * n_q broadcasts are issued before any is waited for: the collectives queue back to back
                pending += [self.dist.broadcast(b, src=owner, async_op=True) for b in bufs]
        return pending

    def begin_packed(self, local_first, n_per_rank, base=0):
        """The block exchange in TWO collectives whatever the block size: every rank contributes its local slots local_first ..
        local_first + n_per_rank - 1 (contiguous in its set, so the payload is a view, not a copy) and one all-gather per payload
        region drops them into the contiguous query buffers base .. base + world * n_per_rank - 1 of every rank: query j of rank r
        lands in buffer base + r * n_per_rank + j. The shape of fastcar's outer loop (fastcar/FC_Runner.cpp:585-597) turned
        around: the database shards stay put, each step's query chunk is assembled from all of them.

        backend.export_block(local_first, n) -> list of tensors [n, row_bytes] (this rank's rows)
        backend.block_buffers(base, n_rows)  -> list of tensors [n_rows, row_bytes] the gathered rows land in"""
        srcs = self.backend.export_block(local_first, n_per_rank)
        dsts = self.backend.block_buffers(base, self.plan.world * n_per_rank)
        if self.plan.world > 1:
            return [self.dist.all_gather_into_tensor(d, s_, async_op=True) for d, s_ in zip(dsts, srcs)]
        for d, s_ in zip(dsts, srcs):
            d.copy_(s_)
        return []

    def packed_globals(self, local_first, n_per_rank):
        """global indices of the queries of a packed block, in buffer order"""
        return [self.plan.global_index(r, local_first + j) for r in range(self.plan.world) for j in range(n_per_rank)]

    def finish(self, pending, n, base=0):
        for w in pending:
            w.wait()
        self.backend.import_queries(n, base)

    class _Totals:
        """the per-query close counts of a block, all ranks summed: the all-gather was only ISSUED by score(defer=True); total() waits
        for it and brings the sums to the host"""
        def __init__(self, work, out, world):
            self.work, self.out, self.world, self.value = work, out, world, None

        def total(self):
            if self.value is None:
                if self.work is not None:
                    self.work.wait()
                self.value = self.out.view(self.world, -1).sum(dim=0).cpu().numpy()
            return self.value

    def score(self, n, base=0, defer=False):
        """-> (close flags of this rank's shard, per-query close counts over all ranks). defer=True: the counts' all-gather is issued
        and a _Totals handle returned in their place -- the step does not wait for a collective it needs nothing from"""
        import torch
        close = self.backend.score_block(n, base)
        # a backend may hand back (flags, per-query counts): a block of 1 024 queries x 12 500 candidates is 12.8 MB of flags, 6-10 ms of
        # numpy's uint8 row sums against a step of ~15 ms on an 8-rank run, while the library adds them up on the device
        if isinstance(close, tuple):
            close, row_counts = close
        else:
            row_counts = np.einsum("ij->i", close, dtype=np.uint64)
        counts = torch.tensor(np.asarray(row_counts, dtype=np.float64), dtype=torch.float64, device=self.device)
        if self.plan.world > 1:
            out = torch.zeros(self.plan.world * counts.numel(), dtype=torch.float64, device=self.device)
            work = self.dist.all_gather_into_tensor(out, counts, async_op=True)
            totals = self._Totals(work, out, self.plan.world)
        else:
            totals = self._Totals(None, counts, 1)
        return close, (totals if defer else totals.total())

    def score_block(self, query_globals):
        n = len(query_globals)
        self.finish(self.begin(query_globals), n)
        return self.score(n)


class ReplicatedRows:
    """The all-pairs job sharded by QUERY ROWS with the candidates replicated: the shape of fastcar's outer loop as the reference runs
    it -- a worker takes a chunk of one side and ALL of the other (fastcar/FC_Runner.cpp:585-597 over work() :426-471). Every rank
    generates (or reads) only its own shard of the sequences; ONE set-up exchange (an all-gather per payload region) gives every
    rank all of them, each rank builds the whole candidate set locally, and from then on a step needs no collective on its data
    path: rank r scores rows r * n / N .. (r + 1) * n / N - 1 of the step's query list against every candidate. Only the per-query
    close counts travel (one small all-gather per step, issued asynchronously and read a step later).

    backend.shard_payload(n_pad)   -> list of tensors [n_pad, row_bytes]: this rank's sequences in local order (rows past its count: padding)
    backend.gather_buffers(n_rows) -> list of tensors [n_rows, row_bytes] the gathered rows land in (n_rows = world * n_pad)
    backend.import_rows(rows)      -> rows[g] = gathered row of global sequence g; builds the resident set of ALL sequences, slot = g
    backend.score_rows(globals_)   -> per-query close counts (np array, len(globals_)) of those query rows against ALL candidates
    """

    def __init__(self, dist, plan, backend, rank, device="cpu"):
        self.dist, self.plan, self.backend, self.rank, self.device = dist, plan, backend, rank, device

    def replicate(self):
        """the set-up exchange -> rows (np.int64, one per global sequence: its row in the gathered buffers)"""
        return ShardedCentres(self.dist, self.plan, _RowsAsCentres(self.backend), self.rank, self.device).gather()

    def rows_of(self, query_globals, rank=None):
        """the slice of a step's query list a rank scores (equal shares, rank-major)"""
        rank = self.rank if rank is None else rank
        n, world = len(query_globals), self.plan.world
        if n % world:
            raise ValueError("a step's queries (%d) must divide evenly over %d ranks" % (n, world))
        per = n // world
        return query_globals[rank * per:(rank + 1) * per]

    class _Counts:
        """the per-query close counts of a step in the order of its query list; the all-gather was only issued by score(defer=True)"""
        def __init__(self, work, out):
            self.work, self.out, self.value = work, out, None

        def total(self):
            if self.value is None:
                if self.work is not None:
                    self.work.wait()
                self.value = self.out.cpu().numpy()
            return self.value

    def score(self, query_globals, defer=False):
        """this rank's rows of the step against every candidate -> the step's per-query close counts (all ranks, query-list order), or
        with defer=True a handle whose total() waits for the counts' all-gather"""
        import torch
        mine = self.rows_of(query_globals)
        counts = torch.as_tensor(np.asarray(self.backend.score_rows(mine), dtype=np.int64), device=self.device)
        if self.plan.world > 1:
            out = torch.zeros(self.plan.world * counts.numel(), dtype=torch.int64, device=self.device)
            work = self.dist.all_gather_into_tensor(out, counts, async_op=True)
            handle = self._Counts(work, out)
        else:
            handle = self._Counts(None, counts)
        return handle if defer else handle.total()


class _RowsAsCentres:
    """adapter: ReplicatedRows' backend seen through the names ShardedCentres.gather() calls (the exchange is the same: every rank's
    rows, padded to the longest share, all-gathered per payload region, then a global-index -> gathered-row map)"""
    def __init__(self, backend):
        self.backend = backend

    def centre_payload(self, n_pad):
        return self.backend.shard_payload(n_pad)

    def gather_buffers(self, n_rows):
        return self.backend.gather_buffers(n_rows)

    def import_centres(self, rows):
        return self.backend.import_rows(rows)


def device_tensors(hist_set, n_slots):
    """torch uint8 views [n_slots, slot_bytes] / [n_slots, scalar_bytes] over a dense set's device memory
    (msc_hist_set_device_view), so that RCCL collectives read and write histogram slots in place.
    torch must have been imported BEFORE the library was loaded (one HIP runtime per process: torch brings its own)."""
    import torch
    bins_ptr, slot_bytes, scal_ptr, scal_bytes = hist_set.device_view()

    class _View:
        def __init__(self, ptr, nbytes):
            self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
    bins = torch.as_tensor(_View(bins_ptr, slot_bytes * n_slots), device="cuda").view(n_slots, slot_bytes)
    scal = torch.as_tensor(_View(scal_ptr, scal_bytes * n_slots), device="cuda").view(n_slots, scal_bytes)
    return bins, scal


class ShardedCentres:
    """The update-round exchange of SURVEY 8(e): centre j (position in `part`, cluster/ClusterFactory.cpp:386) lives on rank
    plan.owner(j); one ALL-GATHER per payload region replicates every centre histogram on every rank, after which the
    Trainer::merge scan over neighbouring centres (cluster/ClusterFactory.cpp:386-388) and the filter neighbourhoods of
    mean_shift_update are local and give every rank the single-process answer.

    backend.centre_payload(n_pad)  -> list of tensors [n_pad, row_bytes]: this rank's centres in local order, rows past its
                                      own count are padding (never read back)
    backend.gather_buffers(n_rows) -> list of tensors [n_rows, row_bytes] the gathered rows land in (n_rows = world * n_pad)
    backend.import_centres(rows)   -> rows[j] = gathered row of global centre j; called once the collectives have completed
    """

    def __init__(self, dist, plan, backend, rank, device="cpu"):
        self.dist, self.plan, self.backend, self.rank, self.device = dist, plan, backend, rank, device

    def gather(self):
        """-> rows (np.int64, one per global centre); plan.n_total is the number of centres of this round"""
        world = self.plan.world
        n_pad = max(self.plan.local_count(r) for r in range(world))
        payload = self.backend.centre_payload(n_pad)
        out = self.backend.gather_buffers(world * n_pad)
        if world > 1:
            pending = [self.dist.all_gather_into_tensor(o, p, async_op=True) for o, p in zip(out, payload)]
            for w in pending:
                w.wait()
        else:
            for o, p in zip(out, payload):
                o.copy_(p)
        rows = np.array([self.plan.owner(j) * n_pad + self.plan.local(j) for j in range(self.plan.n_total)], dtype=np.int64)
        self.backend.import_centres(rows)
        return rows
