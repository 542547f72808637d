"""Mean-shift clustering over one GPU PER RANK (SURVEY.md 8(e) + 8(f1); BASELINE cfg3 / cfg5 shape).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         -m meshclust2_amd.cluster input.fa --recover weights.txt [--id 0.9] [--kmer K] [--datatype 8|16|32|64] [--output out.clstr]

Division of labour
  - the clustering LOGIC (ClusterFactory::MS / accumulate / mean_shift_update / merge / print_output, bvec; cluster/ClusterFactory.cpp
    :288-435,553-656, cluster/bvec.cpp) is the host C++ of libmsc_driver.so (host/msc_driver.hpp), the same code msc_cluster runs on one
    GPU. EVERY rank runs it on replicated bookkeeping (length bins, marks, cluster lists): given the same operator results it takes
    the same decisions, so no bookkeeping is ever exchanged.
  - the POINT histograms are sharded: point p (position in the input) lives on rank plan.owner(p), block-cyclic in blocks of 1000
    (one bvec bin, cluster/CRunner.cpp:585); every rank scores only its own candidates with libmeshclust2_hip.so.
  - the CENTRE histograms are replicated: a centre is a clone of a point (cluster/Center.h:13-40), created from the query the
    owner has just broadcast, so the merge scan (cluster/ClusterFactory.cpp:383-401) and the filter of an update round read
    local centres on every rank.
Exchanges per operator (DistributedBackend), RCCL over xGMI when the ranks own GPUs ("nccl" IS RCCL on ROCm), gloo on CPU:
  get_close   broadcast of the query slot (bins + scalar record) from its owner unless it is already resident; all-gather of one
              (n_close, best_sim, best window position) record per rank, folded with the reference's first-maximum rule; all-gather
              of the close window positions (skipped when nothing is close)
  closest     (get_mean / Trainer::closest) one all-gather per payload region collects the member histograms on every rank, then
              the local msc_mean_nearest gives every rank the same answer
  filter      all-gather of the surviving list positions
  centre_new / centre_set   the broadcast of get_close, nothing else
  merge       none (centres are replicated)
Not batched yet: the update round goes centre by centre (the driver's serial order), one filter + closest + centre_set each.

The rank-local work sits behind an Engine: GpuEngine (this file) is libmeshclust2_hip.so through meshclust2_amd.api and torch views
of its device memory; tests/test_cluster_gloo.py plugs in the CPU oracle (test infrastructure) to run the N > 1 path without a GPU.
"""
import argparse
import os
import sys

import numpy as np

from . import _driver, shard


# ------------------------------------------------------------------ collectives
class Exchange:
    """The few collectives the backend needs, over torch.distributed (or none at all for one rank). stage_cpu: the payload tensors
    live on a GPU but the process group is gloo (two ranks sharing one GPU on a 1-GPU box): stage through host memory."""

    def __init__(self, dist, rank, world, device="cpu", stage_cpu=False):
        self.dist, self.rank, self.world, self.device, self.stage_cpu = dist, rank, world, device, stage_cpu
        self.calls = {"broadcast": 0, "all_gather": 0}

    def broadcast(self, tensors, src):
        if self.world == 1:
            return
        for t in tensors:
            self.calls["broadcast"] += 1
            if self.stage_cpu:
                h = t.cpu()
                self.dist.broadcast(h, src=src)
                if self.rank != src:
                    t.copy_(h)
            else:
                self.dist.broadcast(t, src=src)

    def gather_rows(self, outs, ins):
        """outs[i] ([world * n, bytes]) <- every rank's ins[i] ([n, bytes]), rank-major"""
        for o, i in zip(outs, ins):
            if self.world == 1:
                o.copy_(i)
                continue
            self.calls["all_gather"] += 1
            if self.stage_cpu:
                ho, hi = o.cpu(), i.cpu().contiguous()
                self.dist.all_gather_into_tensor(ho, hi)
                o.copy_(ho)
            else:
                self.dist.all_gather_into_tensor(o, i.contiguous())

    def gather_vec(self, vec, dtype=np.float64):
        """vec (1-D, same length on every rank) -> [world, len]"""
        import torch
        v = np.ascontiguousarray(vec, dtype=dtype)
        if self.world == 1:
            return v.reshape(1, -1)
        self.calls["all_gather"] += 1
        dev = "cpu" if self.stage_cpu else self.device
        t = torch.from_numpy(v).to(dev)
        out = torch.empty(self.world * t.numel(), dtype=t.dtype, device=dev)          # flat: rank r's vector at [r * len, (r + 1) * len)
        self.dist.all_gather_into_tensor(out, t)
        return out.cpu().numpy().reshape(self.world, -1)

    def gather_lists(self, mine, counts=None):
        """variable-length int64 lists -> their concatenation in rank order; counts[r] (if every rank already knows them) saves a round"""
        mine = np.ascontiguousarray(mine, dtype=np.int64)
        if self.world == 1:
            return mine
        if counts is None:
            counts = self.gather_vec(np.array([mine.size]), np.int64)[:, 0]
        top = int(max(counts))
        if top == 0:
            return np.zeros(0, dtype=np.int64)
        pad = np.full(top, -1, dtype=np.int64)
        pad[:mine.size] = mine
        allv = self.gather_vec(pad, np.int64)
        return np.concatenate([allv[r, :int(counts[r])] for r in range(self.world)])


# ------------------------------------------------------------------ the seven operators, sharded
class DistributedBackend:
    """msc::ClusterBackend (host/msc_driver.hpp) over sharded points: see the module docstring. `engine` does the rank-local work:
         n_local                                   points of this rank (local slot = plan.local(point))
         point_payload(local) -> [tensors]         the payload of one local point (views)
         query_payload() -> [tensors]              where a broadcast query lands;  query_ready() once it has
         get_close(local_slots) -> (flags, best position in local_slots or -1, best_sim)        against the resident query
         centre_from_query(centre or None, clone) -> centre     Center(c->clone()) / center->set(*next) from the resident query
         filter(centre, local_slots) -> keep;   merge(centres, current, begin, last) -> best;   merge_all(centres, delta) -> best[n] (optional)
         member_payload(local_slots, n_pad) -> [tensors [n_pad, bytes]];   scratch_payload(n_rows) -> [tensors [n_rows, bytes]]
         mean_nearest(rows) -> position            over scratch rows, in the order given, once scratch_ready(n_rows) was called"""

    def __init__(self, engine, plan, exchange, rank):
        self.e, self.plan, self.x, self.rank = engine, plan, exchange, rank
        self.resident = -1          # point whose histogram sits in every rank's query slot
        # MSC_CLUSTER_TRACE=path: rank 0 writes one line per operator call (arguments and result, hashed where long), so that runs with
        # different numbers of ranks can be compared call by call -- the driver's logic is the same, the first differing line is the bug
        self.trace = open(os.environ["MSC_CLUSTER_TRACE"], "w") if rank == 0 and os.environ.get("MSC_CLUSTER_TRACE") else None

    def _t(self, what, *vals):
        if self.trace is not None:
            import zlib
            out = []
            for v in vals:
                a = np.asarray(v)
                out.append(str(v) if a.ndim == 0 else "%d:%08x" % (a.size, zlib.crc32(np.ascontiguousarray(a).tobytes())))
            self.trace.write(what + " " + " ".join(out) + "\n")
            self.trace.flush()

    def _owners(self, points):
        return (points.astype(np.int64) // self.plan.block) % self.plan.world

    def _locals(self, points):
        p = points.astype(np.int64)
        b = p // self.plan.block
        return ((b // self.plan.world) * self.plan.block + p % self.plan.block).astype(np.uint32)

    def _make_resident(self, point):
        if self.resident == point:
            return
        owner = self.plan.owner(point)
        bufs = self.e.query_payload()
        if self.rank == owner:
            for dst, src in zip(bufs, self.e.point_payload(self.plan.local(point))):
                dst.copy_(src)
        self.x.broadcast(bufs, owner)
        self.e.query_ready()
        self.resident = point

    def get_close(self, q, window):
        self._make_resident(q)
        mine = np.flatnonzero(self._owners(window) == self.rank)            # window positions this rank scores, in window order
        flags_l, best_l, sim_l = self.e.get_close(self._locals(window[mine]))
        close_pos = mine[np.flatnonzero(flags_l)]
        rec = self.x.gather_vec([close_pos.size, sim_l if best_l >= 0 else -1.0, mine[best_l] if best_l >= 0 else -1])
        n_close = rec[:, 0].astype(np.int64)
        # Trainer::get_close keeps the FIRST maximum in window order (strict '>' at one thread, cluster/Trainer.cpp:26-37,59)
        best_pos, best_sim = -1, -1.0
        for r in range(self.x.world):
            p, s_ = int(rec[r, 2]), float(rec[r, 1])
            if p >= 0 and (best_pos < 0 or s_ > best_sim or (s_ == best_sim and p < best_pos)):
                best_pos, best_sim = p, s_
        flags = np.zeros(window.size, dtype=np.uint8)
        if n_close.sum():
            flags[self.x.gather_lists(close_pos, n_close)] = 1
        self._t("get_close", q, window.astype(np.int64), flags, best_pos, best_sim)
        return flags, best_pos, int(n_close.sum()) == 0

    def closest(self, members):
        if members.size == 1:
            return 0
        owners = self._owners(members)
        counts = np.bincount(owners, minlength=self.x.world)
        n_pad = int(counts.max())
        mine = np.flatnonzero(owners == self.rank)
        self.x.gather_rows(self.e.scratch_payload(self.x.world * n_pad), self.e.member_payload(self._locals(members[mine]), n_pad))
        self.e.scratch_ready(self.x.world * n_pad)
        # member i is the k-th member of its owner r: gathered row r * n_pad + k
        seen = np.zeros(self.x.world, dtype=np.int64)
        rows = np.zeros(members.size, dtype=np.uint32)
        for i, r in enumerate(owners):
            rows[i] = r * n_pad + seen[r]
            seen[r] += 1
        res = self.e.mean_nearest(rows)
        self._t("closest", members.astype(np.int64), res)
        return res

    def centre_new(self, point):
        self._make_resident(point)
        self._t("centre_new", point)
        return self.e.centre_from_query(None, clone=True)

    def centre_set(self, centre, point):
        self._make_resident(point)
        self._t("centre_set", centre, point)
        self.e.centre_from_query(centre, clone=False)

    def filter(self, centre, points):
        mine = np.flatnonzero(self._owners(points) == self.rank)
        keep_l = self.e.filter(centre, self._locals(points[mine]))
        keep = np.zeros(points.size, dtype=np.uint8)
        keep[self.x.gather_lists(mine[np.flatnonzero(keep_l)])] = 1
        self._t("filter", centre, points.astype(np.int64), keep)
        return keep

    def merge(self, centres, current, begin, last):
        res = self.e.merge(centres, current, begin, last)
        self._t("merge", np.asarray(centres, dtype=np.int64), current, begin, last, res)
        return res

    def merge_all(self, centres, delta):
        res = self.e.merge_all(centres, delta)
        self._t("merge_all", np.asarray(centres, dtype=np.int64), delta, np.asarray(res, dtype=np.int64))
        return res


# ------------------------------------------------------------------ rank-local work on the GPU
class GpuEngine:
    """The rank's shard in a dense msc_hist_set (slot n_local = the query slot), a replicated centre store and a scratch set for the
    gathered members of get_mean, all reached through the C ABI; payloads are torch views of the sets' device memory, so the
    collectives read and write histogram slots in place. torch must be imported before the library is loaded."""

    def __init__(self, api, ctx, k, dtype, seqs, feat, cutoff):
        self.api, self.ctx, self.k, self.dtype, self.cutoff = api, ctx, k, dtype, cutoff
        self.n_local = len(seqs)
        self.points = api.HistogramSet(ctx, k, dtype, self.n_local + 1)
        for off in range(0, self.n_local, 8192):
            self.points.build(seqs[off:off + 8192], first_slot=off)
        self.trn = api.Trainer(ctx, feat, cutoff)
        self.p_bins, self.p_scal = shard.device_tensors(self.points, self.n_local + 1)
        self.centres = api.HistogramSet(ctx, k, dtype, 256)
        self.n_centres = 0
        self.scratch = None
        self.s_rows = 0

    def lengths(self):
        return self.points.lengths(0, self.n_local).astype(np.int64)

    def point_payload(self, local):
        return [self.p_bins[local], self.p_scal[local]]

    def query_payload(self):
        return [self.p_bins[self.n_local], self.p_scal[self.n_local]]

    def _sync(self):
        import torch
        torch.cuda.synchronize()

    def query_ready(self):
        self._sync()
        self.points.import_done(self.n_local, 1)

    def get_close(self, local_slots):
        flags, pos, sim, _ = self.trn.get_close(self.points, local_slots, self.points, self.n_local)
        return flags, pos, sim

    def centre_from_query(self, centre, clone):
        if clone:
            if self.n_centres == self.centres.capacity:          # relocate into a store twice the size (exact copies: stale mags survive)
                bigger = self.api.HistogramSet(self.ctx, self.k, self.dtype, 2 * self.centres.capacity)
                every = np.arange(self.n_centres, dtype=np.uint32)
                bigger.copy_batch(every, self.centres, every)
                self.centres.close()
                self.centres = bigger
            centre = self.n_centres
            self.n_centres += 1
            self.centres.clone_from(centre, self.points, self.n_local)
        else:
            self.centres.assign_from(centre, self.points, self.n_local)
        # clone / set queue their copies on the library's stream and return; the next broadcast overwrites the query slot on torch's
        # stream, which knows nothing of that queue -- without this the centre could receive the NEXT query's histogram
        self.ctx.synchronize()
        return centre

    def filter(self, centre, local_slots):
        return self.trn.filter(self.centres, centre, self.points, local_slots) if local_slots.size else np.zeros(0, dtype=np.uint8)

    def merge(self, centres, current, begin, last):
        return self.trn.merge(self.centres, centres, current, begin, last)

    def merge_all(self, centres, delta):
        return self.trn.merge_all(self.centres, centres, delta)

    def member_payload(self, local_slots, n_pad):
        import torch
        idx = torch.from_numpy(local_slots.astype(np.int64)).to(self.p_bins.device)
        out = []
        for region in (self.p_bins, self.p_scal):
            stage = torch.zeros((n_pad, region.shape[1]), dtype=torch.uint8, device=region.device)
            if local_slots.size:
                stage[:local_slots.size] = region.index_select(0, idx)
            out.append(stage)
        return out

    def scratch_payload(self, n_rows):
        if n_rows > self.s_rows:
            if self.scratch is not None:
                self.scratch.close()
            self.s_rows = max(64, 2 * n_rows)
            self.scratch = self.api.HistogramSet(self.ctx, self.k, self.dtype, self.s_rows)
            self.s_bins, self.s_scal = shard.device_tensors(self.scratch, self.s_rows)
        return [self.s_bins[:n_rows], self.s_scal[:n_rows]]

    def scratch_ready(self, n_rows):
        self._sync()
        self.scratch.import_done(0, n_rows)

    def mean_nearest(self, rows):
        return self.api.mean_nearest(self.ctx, self.scratch, rows)[0]


# ------------------------------------------------------------------ FASTA (nonltr/ChromListMaker.cpp:24-48,117-165)
def read_fasta(path):
    """-> (headers incl. '>', sequences as bytes): CR / LF / CRLF line ends, lines that start with a blank are skipped, text in
    front of the first header is dropped"""
    headers, seqs = [], []
    for line in open(path, "rb").read().replace(b"\r\n", b"\n").replace(b"\r", b"\n").split(b"\n"):
        if line[:1] == b">":
            headers.append(line.decode(errors="replace"))
            seqs.append([])
        elif line[:1] in (b" ", b"\t"):
            continue
        elif headers:
            seqs[-1].append(line)
    return headers, [b"".join(s_) for s_ in seqs]


def cluster(engine_factory, plan, exchange, rank, headers, similarity, delta=5, iterations=15, output=None, log=None):
    """Runs the mean-shift logic on this rank. engine_factory() -> engine over this rank's points (plan.local_globals(rank), in local
    order) with .lengths(); only rank 0 writes `output`. Returns the engine."""
    engine = engine_factory()
    # effective lengths of ALL points: every rank knows its own, one all-gather spreads them
    n_pad = max(plan.local_count(r) for r in range(plan.world))
    mine = np.full(n_pad, -1, dtype=np.int64)
    mine[:engine.n_local] = engine.lengths()
    allv = exchange.gather_vec(mine, np.int64)
    lengths = np.zeros(plan.n_total, dtype=np.int64)
    for r in range(plan.world):
        g = plan.local_globals(r)
        lengths[g] = allv[r, :g.size]
    backend = DistributedBackend(engine, plan, exchange, rank)
    _driver.run(backend, headers, lengths, similarity, delta, iterations, output=output if rank == 0 else None,
                log=log if log is not None else (None if rank == 0 else os.devnull), batch_update=True)
    return engine


def main(argv=None):
    ap = argparse.ArgumentParser(description="mean-shift clustering, one GPU per rank (see the module docstring)")
    ap.add_argument("fasta")
    ap.add_argument("--recover", "-r", required=True, help="weights file (meshclust2 --dump / weights.txt, or msc_cluster's)")
    ap.add_argument("--id", type=float, default=None)
    ap.add_argument("--kmer", "-k", type=int, default=None)
    ap.add_argument("--datatype", default=None)
    ap.add_argument("--output", "-o", default="output.clstr")
    ap.add_argument("--delta", "-d", type=int, default=5)
    ap.add_argument("--iterations", "-i", type=int, default=15)
    args = ap.parse_args(argv)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch          # before the library: one HIP runtime per process
    dist = None
    one_gpu = os.environ.get("MSC_BENCH_ONE_GPU") == "1"      # two ranks share device 0 over gloo: a smoke test of the exchange on a 1-GPU box
    backend_name = os.environ.get("MSC_BENCH_BACKEND", "nccl")
    if one_gpu:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if backend_name == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend_name)
    from . import api
    text = open(args.recover).read()
    head = dict(ln.split(":", 1) for ln in text.splitlines()[:6] if ":" in ln)
    k = args.kmer if args.kmer is not None else int(head["k"])
    sim = args.id if args.id is not None else float(head["ID"])
    dt = args.datatype or head["Datatype"].strip()
    dtype = {"uint8_t": 8, "uint16_t": 16, "uint32_t": 32, "uint64_t": 64}.get(dt) or int(dt)
    headers, seqs = read_fasta(args.fasta)
    plan = shard.ShardPlan(len(seqs), world, block=1000)
    ctx = api.Context(local_rank)
    ctx.set_kernel_timing(False)          # one get_close per step: no per-call event records
    feat = api.Feature.from_text(ctx, text, 0)
    x = Exchange(dist, rank, world, device="cuda", stage_cpu=world > 1 and backend_name != "nccl")
    cluster(lambda: GpuEngine(api, ctx, k, dtype, [seqs[g] for g in plan.local_globals(rank)], feat, sim), plan, x, rank, headers, sim, args.delta,
            args.iterations, output=args.output)
    if rank == 0:
        print("collectives: %s" % x.calls, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
