#!/usr/bin/env python3
"""bench.py -- sequence-pairs/s identity-scored on MI355X (BASELINE.json metric), with the pair kernel's HBM roofline
and the CPU path timed beside it.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--nseq 100000] [--k 9] [--dtype 32] [--queries 64] [--scaling strong|weak]

Workload (BASELINE.json configs[1], "cfg2"): `nseq` synthetic 1 kb sequences IN TOTAL (families of 20, 3 % substitutions,
0.5 % indels, SplitMix64 seed 20260002; sequence g is the same whatever the number of ranks), k = 9, --datatype 32 -> 1 MiB
per histogram, all resident in HBM before the timed region. One STEP = one block of the full pairwise matrix: `queries`
query histograms x ALL histograms (Predictor::close of fastcar's work(), fastcar/FC_Runner.cpp:426-471; the same per-pair
work as cluster/Trainer.cpp:49-52): 9 `fast` statistics, normalise, 4 combos, GLM, close flags copied back to the host.
--mode allpairs (default) scores the block in ONE streaming pass (msc_score_multi -> k_pair_digest_multi: a workgroup fetches
each candidate tile once for 16 queries); --mode get_close issues `queries` separate 1 x M
Trainer::get_close passes (the step-serial shape of the mean-shift accumulate loop, incl. the arg-max reduce).
value = scored pairs per second over the whole job (all ranks).
roofline: algorithmic bytes per launch = (M * ceil(Q / q) + Q) * 4^k * sizeof(T), q = queries served per HBM read of a
candidate tile as msc_last_kernel_info reports it for the kernel that actually ran, divided by the kernel's own launch time
(HIP events on the library's stream).

Multi-GPU (--gpus N, launched by torch.distributed.run): the histograms are sharded by sequence block (block-cyclic, 1000 per
block: meshclust2_amd/shard.py). --scaling strong (default; north_star: ">= 6x 1 -> 8 GPUs on 100k x 1kb"): `nseq` is the TOTAL,
every rank holds nseq / N of them; --scaling weak: `nseq` per GPU. Per step every rank contributes queries / N of its own
histograms and ONE all-gather per payload region (bins, scalar records: 2 collectives per step, RCCL over xGMI) assembles the
step's query block in every rank's query slots -- fastcar's chunked outer loop (fastcar/FC_Runner.cpp:585-597) with the
database shards resident and the query chunk moving; the exchange of step s + 1 runs underneath the kernel of step s. Each
rank scores the block against its shard and the per-query close counts are all-gathered. --mode get_close broadcasts one
query per pass from its owner and all-gathers one 24-byte (n_close, best_sim, best_global_index) record per rank.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_FP4_PEAK_TOPS = 10000.0  # dense FP4 matrix rate (v_mfma_f32_32x32x64_f8f6f4 with E2M1 operands): 4 x the bf16 rate per clock (same guide)
MFMA_I8_PEAK_TOPS = 5000.0  # dense int8 matrix rate: 2 x the bf16 rate per clock (same guide, Matrix cores: bf16 ~2.5 PF dense)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed steps; the default scores 20 x 1024 x 100000 = 2.05e9 pairs (SURVEY 8d: a >= 1e8-pair slice)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nseq", type=int, default=100000, help="sequences in total (--scaling strong) or per GPU (--scaling weak)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--length", type=int, default=1000)
    ap.add_argument("--k", type=int, default=9)
    ap.add_argument("--dtype", type=int, default=32)
    ap.add_argument("--exchange", choices=("sequences", "histograms"), default="sequences",
                    help="what the ranks of an N > 1 all-pairs run send each other per step: the queries' 2-bit packed SEQUENCES (250 bytes for 1 kb; every "
                         "rank builds the block's histograms itself, bit-identical to the owner's) or their HISTOGRAMS as they sit in HBM (4^k x sizeof(T) "
                         "bytes each: SURVEY 8(e)'s exchange -- at 1 MiB per query the ring all-gather takes longer than scoring the block once the pass "
                         "runs on the matrix cores)")
    ap.add_argument("--queries", type=int, default=8192, help="query histograms per step IN TOTAL (a multiple of the number of ranks); the library scores them in blocks of 128 "
                                                             "(64 off the matrix cores), one pass over the candidates each, queued on two streams -- 8 192 leaves each of 8 ranks "
                                                             "eight blocks per step, enough for that pipeline to fill (r04 and before: 1 024)")
    ap.add_argument("--shard", choices=("rows", "candidates"), default="rows",
                    help="how an N > 1 all-pairs run is cut. rows (default): every rank holds ALL candidates (one set-up all-gather of the 2-bit sequences, each "
                         "rank builds the whole set: 103 GiB of 288 at cfg2) and scores queries / N rows of each step against all of them -- fastcar's own cut "
                         "(a worker takes a chunk of one side and all of the other, fastcar/FC_Runner.cpp:585-597); no collective on the data path, only the "
                         "per-query close counts are all-gathered, one step late. candidates: SURVEY 8(e)'s cut -- every rank holds nseq / N candidates and the "
                         "step's query block is all-gathered (r01-r04)")
    ap.add_argument("--mode", choices=("allpairs", "get_close"), default="allpairs")
    ap.add_argument("--layout", choices=("dense", "sparse"), default="dense",
                    help="sparse: the same workload on sorted (bin, value) lists (DESIGN 3b; single GPU only) -- algorithmic bytes are then the list bytes")
    ap.add_argument("--weights", default=None, help="weights file (default: tests/golden/weights_k9_u32.txt, the 9-feature `fast` model of cfg2); e.g. "
                                                    "tests/golden/weights_cfg5_k9.txt for a `--feat slow` model with jensen_shannon")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (0 disables it)")
    ap.add_argument("--cpu-cands", type=int, default=16384, help="candidates per CPU get_close pass: enough for every host thread of a 256-thread box to get 64")
    ap.add_argument("--repeats", type=float, default=0.0, help="fraction of the sequences that carry a 400-base run (a homopolymer or a dinucleotide repeat, alternating): "
                                                                "counts of ~390 / ~196 among counts of 1 .. 8, as ordinary FASTA has them (microsatellites, poly-A tails); 0.01 = VERDICT r03's case")
    ap.add_argument("--no-secondary", dest="secondary", action="store_false", help="skip the three short 1 x M legs behind the timed region (dense get_close, "
                                                                                   "sparse k = 9, sparse k = 13 / 64-bit / 20 kb) that go under `secondary` in the line")
    ap.add_argument("--check", action="store_true", help="add `check` to the line: per timed step, the number of close candidates of every query of the block (all ranks "
                                                         "summed) -- what two runs with different numbers of ranks must agree on")
    ap.add_argument("--check-world", type=int, default=0, help="single-rank run: choose each step's query block as a run over this many ranks would (with --check)")
    return ap.parse_args()


BLOCK = 1000          # sequences per shard block = one bvec bin (cluster/CRunner.cpp:585)


def build_resident_set(api, synth, ctx, args, plan, rank, build=True, tail=None):
    """This rank's shard resident in HBM before the clock starts: global sequence g = plan.global_index(rank, local) -- the same
    sequence whatever the number of ranks. Sequences are generated and packed on the host in chunks of whole shard blocks.
    build=False: only the host side -- the shard's 2-bit rows and 1-mer / length records (--shard rows: the set is built from the
    rows of ALL ranks after the set-up exchange, build_replicated_set)."""
    M = plan.local_count(rank)
    Q = args.queries if tail is None else tail
    sparse_entries = (M + 2 * Q) * (args.length + 64) if args.layout == "sparse" else 0
    hs = api.HistogramSet(ctx, args.k, args.dtype, M + 2 * Q, sparse_entries=sparse_entries) if build else None     # tail slots = two blocks of gathered queries
    seed = 20260002
    fam = 20
    assert BLOCK % fam == 0
    t0 = time.time()
    t_dev = []          # (sequences, seconds) per chunk inside msc_hist_build_packed only: H2D of the 2-bit bases + the build kernels, host packing excluded
    done = 0
    last = None
    # --exchange sequences: this rank's sequences as they go over the wire, 4 bases per byte, each padded to whole bytes, + their 1-mer counts
    # (the family members carry indels: lengths vary by a few bases around --length, so a row has room to spare and column 4 of the
    # second array holds the length)
    lp4 = (args.length + args.length // 8 + 64 + (400 if args.repeats > 0 else 0) + 3) // 4
    period = max(1, int(round(1.0 / args.repeats))) if args.repeats > 0 else 0
    runs = (np.zeros(400, dtype=np.uint8), np.tile(np.array([0, 1], dtype=np.uint8), 200))          # A x 400, AC x 200
    seq_rows = np.zeros((M, lp4), dtype=np.uint8)
    one_rows = np.zeros((M, 5), dtype=np.uint64)
    while done < M:
        n = min(20 * BLOCK, M - done)
        codes = []
        for lb in range(done // BLOCK, (done + n + BLOCK - 1) // BLOCK):          # local blocks of this chunk
            g0 = plan.global_index(rank, lb * BLOCK)
            g1 = min(g0 + BLOCK, plan.n_total)
            # families are generated by template index, so blocks are independent and reproducible
            for t in range(g0 // fam, (g1 + fam - 1) // fam):
                tmpl = synth.template(seed, t, args.length)
                for j in range(fam):
                    if g0 <= t * fam + j < g1:
                        c = synth.member(seed, t, j, tmpl)
                        g = t * fam + j
                        if period and g % period == 7 % period:          # (the construction of tests/ring_variant_check.py: a run spliced in at base 300)
                            c = np.concatenate([c[:300], runs[(g // period) % 2], c[300:]])
                        codes.append(c)
        assert len(codes) == n
        b = synth.pack_batch(codes)
        c4 = np.zeros((n, lp4 * 4), dtype=np.uint8)
        for i, c in enumerate(codes):
            assert c.size <= lp4 * 4
            c4[i, :c.size] = c
        c4 = c4.reshape(n, lp4, 4)
        seq_rows[done:done + n] = c4[:, :, 0] | (c4[:, :, 1] << 2) | (c4[:, :, 2] << 4) | (c4[:, :, 3] << 6)
        one_rows[done:done + n, :4] = b["one_mers"].reshape(n, 4)
        one_rows[done:done + n, 4] = b["eff_len"]
        if build:
            t1 = time.perf_counter()
            hs.build_packed(done, n, b["packed"], b["n_bases"], b["seg_seq"], b["seg_start"], b["seg_end"], b["eff_len"], b["one_mers"])
            ctx.synchronize()
            t_dev.append((n, time.perf_counter() - t1))
        last = (done, n, b)
        done += n
    if not build:
        return None, time.time() - t0, None, (seq_rows, one_rows)
    # Between chunks the host spends ~1 s generating sequences, so every chunk above starts on an idle GPU (and the first ones
    # allocate the library's scratch buffers). Steady state = the last chunk rebuilt in place three times back to back (same
    # bytes into the same slots). Dense only: a sparse set's entry arena is append-only, a rebuilt slot reserves its entries again.
    steady = []
    for _ in range(3 if args.layout == "dense" and M else 0):
        first, n, b = last
        t1 = time.perf_counter()
        hs.build_packed(first, n, b["packed"], b["n_bases"], b["seg_seq"], b["seg_start"], b["seg_end"], b["eff_len"], b["one_mers"])
        ctx.synchronize()
        steady.append(n / (time.perf_counter() - t1))
    rate_all = sum(n_ for n_, _ in t_dev) / sum(s_ for _, s_ in t_dev)
    return hs, time.time() - t0, {"all": rate_all, "median_chunk": max(steady) if steady else rate_all}, (seq_rows, one_rows)


def build_replicated_set(api, shard, torch, dist, ctx, args, plan, rank, seq_rows, one_rows):
    """--shard rows, N > 1: the set-up exchange and the build of ALL candidates on this rank. Every rank contributes the 2-bit rows
    and the (1-mer counts, length) records of the sequences it generated; one all-gather per region (RCCL, device buffers) lands
    them on every rank, a device gather puts them in global order (slot = global sequence index, as on one GPU), and the histograms
    are built straight from that device buffer (msc_hist_build_packed_dev) in chunks of 20 000. -> (set, seconds, sequences per second
    of the build calls)"""
    t0 = time.time()
    n_total, lp4 = plan.n_total, seq_rows.shape[1]
    hs = api.HistogramSet(ctx, args.k, args.dtype, n_total)
    state = {}

    class Backend:
        def shard_payload(self, n_pad):
            a = torch.zeros((n_pad, lp4), dtype=torch.uint8, device="cuda")
            a[:seq_rows.shape[0]] = torch.from_numpy(seq_rows).cuda()
            b = torch.zeros((n_pad, 5), dtype=torch.int64, device="cuda")
            b[:one_rows.shape[0]] = torch.from_numpy(one_rows.view(np.int64)).cuda()
            return [a, b]

        def gather_buffers(self, n_rows):
            state["seq"] = torch.zeros((n_rows, lp4), dtype=torch.uint8, device="cuda")
            state["one"] = torch.zeros((n_rows, 5), dtype=torch.int64, device="cuda")
            return [state["seq"], state["one"]]

        def import_rows(self, rows):
            idx = torch.from_numpy(rows).cuda()
            state["seq"] = state["seq"].index_select(0, idx).contiguous()
            state["meta"] = state["one"].index_select(0, idx).cpu().numpy().view(np.uint64)
            torch.cuda.synchronize()          # the library reads the buffer on its own stream (msc_hist_build_packed_dev: the bytes must be complete)

        def score_rows(self, globals_):
            raise NotImplementedError

    rr = shard.ReplicatedRows(dist, plan, Backend(), rank, device="cuda")
    rr.replicate()
    meta, stride = state["meta"], lp4 * 4
    t_dev = []
    for done in range(0, n_total, 20000):
        n = min(20000, n_total - done)
        lens = meta[done:done + n, 4].copy()
        starts = np.arange(n, dtype=np.uint64) * np.uint64(stride)
        t1 = time.perf_counter()
        hs.build_packed_dev(done, n, state["seq"][done:].data_ptr(), n * stride, np.arange(n, dtype=np.uint32), starts, starts + lens - np.uint64(1), lens,
                            np.ascontiguousarray(meta[done:done + n, :4]).reshape(-1))
        ctx.synchronize()
        t_dev.append((n, time.perf_counter() - t1))
    rate = sum(n_ for n_, _ in t_dev) / sum(s_ for _, s_ in t_dev)
    # the gathered rows stay alive for the secondary-free N > 1 run only as long as this frame: the set owns its histograms
    return hs, time.time() - t0, {"all": rate, "median_chunk": max(n_ / s_ for n_, s_ in t_dev)}, rr


def cpu_baseline(args, synth, weights_text, weights_path):
    """The CPU path on this box's host cores, bounded sample of the same workload. Prefers the real reference
    (oracle/_ref, travels as a prebuilt .so) and falls back to the C port (oracle/libmsc_oracle.so)."""
    from concurrent.futures import ThreadPoolExecutor
    cores = os.cpu_count() or 1
    n = args.cpu_cands + 1
    seqs, _ = synth.families(20260002, n, args.length)
    out = {"unit": "pairs/s", "cores": cores}
    try:
        from oracle import ref_py
        if not ref_py.available():
            raise ImportError("no oracle/_ref")
        ref_py.lib().ref_set_threads(cores)
        with ThreadPoolExecutor(min(cores, 64)) as ex:
            pts = list(ex.map(lambda s: ref_py.Point(args.dtype, s, args.k), seqs))
        model = ref_py.Model(args.dtype, weights_path)
        kind = "reference"

        def one_pass(q):
            return model.get_close(0.9, pts[q], pts[:q] + pts[q + 1:])
    except Exception as e:          # noqa: BLE001 -- any failure of the optional reference build falls back to the port
        from oracle import oracle_py
        oracle_py.lib().orc_set_threads(cores)
        with ThreadPoolExecutor(min(cores, 64)) as ex:
            hs = list(ex.map(lambda s: oracle_py.hist(s, args.k, args.dtype), seqs))
        pred = oracle_py.predictor(weights_text)
        kind = "port"
        out["note"] = "reference .so unavailable (%s)" % type(e).__name__

        def one_pass(q):
            return oracle_py.get_close(pred, 0.9, hs[q], hs[:q] + hs[q + 1:], omp=True)
    one_pass(0)         # warm
    t0 = time.time()
    passes = 0
    while time.time() - t0 < args.cpu_seconds and passes < 64:
        one_pass(passes % n)
        passes += 1
    dt = time.time() - t0
    pairs = passes * (n - 1)
    out.update(value=pairs / dt, kind=kind,
               sample="%d get_close passes x %d candidates (k=%d, u%d, 1 kb), OpenMP over candidates on %d threads, %.1f s"
                      % (passes, n - 1, args.k, args.dtype, cores, dt))
    return out


def get_close_leg(api, ctx, trn, hs, M, passes, hist_bytes, what, rank_bytes=None, pmc_key=None):
    """`passes` 1 x M passes of Trainer::get_close (cluster/Trainer.cpp:23-71: the loop the clustering runs) over a resident set; the
    streaming kernel timed by the library's HIP events on its stream, the leg by the wall clock -> one row of `secondary`"""
    for j in range(3):          # (the third pass over an unchanged set builds its rank lists: untimed)
        trn.get_close(hs, None, hs, (j * 7919 + 1) % M, m=M)
    ctx.synchronize()
    ms, launches = [], []
    t0 = time.perf_counter()
    for j in range(passes):
        trn.get_close(hs, None, hs, (j * 7919 + 3) % M, m=M)
        ms.append(ctx.last_kernel_ms()[0])
        launches.append(ctx.last_kernel_launches())
    ctx.synchronize()
    dt = time.perf_counter() - t0
    kernel, _ = ctx.last_kernel_info()
    n_launch = max(int(np.sum(launches)), 1)
    avg = float(np.sum(ms)) / n_launch
    if kernel.startswith("k_pair_ranks_1xm") and rank_bytes:          # the pass over rank lists reads 4 bytes per k-mer of a candidate
        hist_bytes = rank_bytes
    alg = (M + 1) * hist_bytes * len(ms) // n_launch
    ach = alg / (avg * 1e-3) / 1e9
    traffic = None          # HBM bytes per launch by the counters, where a committed pass of the same shape and kernel exists (profiles/*_pmc_hbm.json)
    if pmc_key and kernel.startswith("k_pair_ranks_1xm"):
        traffic, _ = pmc_traffic("k_pair_ranks_1xm", pmc_key)
    return {"workload": what, "metric": "sequence-pairs/sec identity-scored, 1 x M get_close passes", "value": passes * M / dt, "unit": "pairs/s", "passes": passes,
            "candidates_per_pass": M, "ms_per_pass": dt / passes * 1e3,
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_ms": avg, "algorithmic_bytes_per_launch": alg, "launches_timed": n_launch}}


def secondary_legs(api, synth, ctx, args, hs, M, seq_rows, one_rows, wtext):
    """Four short legs behind the timed region, same process, same resident data where possible: the 1 x M pass the
    clustering loop runs -- over the dense 32-bit set, over the sparse layout of the same sequences (k = 9), and over sparse 64-bit lists
    of 8 000 x 20 kb sequences at k = 13 (BASELINE cfg4's shape). Algorithmic bytes per pass: SURVEY 8(d)'s 4^k sizeof(T) per candidate
    for the dense set; 8 bytes per stored bin of the candidate's list for the merge kernels, 4 bytes per k-mer for k_pair_ranks_1xm."""
    rows = []
    feat = api.Feature.from_text(ctx, wtext, 0)
    trn = api.Trainer(ctx, feat, 0.9)
    what = "the resident set of the main leg: %d x %d bp, k=%d, datatype=%d, dense" % (M, args.length, args.k, args.dtype)
    # the streaming kernel over the bins (SURVEY 8(d)'s 1 x M shape), then what the library does by default since r04: the same pass
    # over the dense set's sparse mirror (8 bytes per counted k-mer)
    ctx.set_mirror_pass(False)
    rows.append(get_close_leg(api, ctx, trn, hs, M, 20, (4 ** args.k) * (args.dtype // 8), what + ", msc_set_mirror_pass(0): the streaming kernel over the bins"))
    ctx.set_mirror_pass(True)
    trn.get_close(hs, None, hs, 1, m=M)          # (builds the mirror: untimed)
    ent0 = int(8 * np.mean([hs.entries(i) for i in range(0, M, max(1, M // 500))]))
    rank_bytes = 4 * (args.length - args.k + 1)
    pmc_key = "get_close over rank lists,nseq=%d,len=%d,k=%d" % (M, args.length, args.k)          # (profiles/r05_ranks_1xm_pmc_hbm.json)
    rows.append(get_close_leg(api, ctx, trn, hs, M, 20, ent0, what + ", default: the lists of its sparse mirror (rank lists up to k = 9)", rank_bytes, pmc_key))
    # the same sequences on the sparse layout, rebuilt from the 2-bit rows kept for the exchange
    stride = seq_rows.shape[1] * 4
    sp = api.HistogramSet(ctx, args.k, args.dtype, M, sparse_entries=M * (args.length + 64 + (400 if args.repeats > 0 else 0)))
    for done in range(0, M, 20000):
        n = min(20000, M - done)
        lens = one_rows[done:done + n, 4].astype(np.uint64)
        starts = np.arange(n, dtype=np.uint64) * np.uint64(stride)
        sp.build_packed(done, n, seq_rows[done:done + n].reshape(-1), n * stride, np.arange(n, dtype=np.uint32), starts, starts + lens - np.uint64(1), lens,
                        np.ascontiguousarray(one_rows[done:done + n, :4]).reshape(-1))
    ent = int(8 * np.mean([sp.entries(i) for i in range(0, M, max(1, M // 500))]))
    rows.append(get_close_leg(api, ctx, trn, sp, M, 20, ent, "the same %d sequences on the sparse layout (sorted (bin, count) lists; rank lists up to k = 9), k=%d" % (M, args.k), rank_bytes, pmc_key))
    del sp
    # BASELINE cfg4's shape: 20 kb sequences at k = 13, 64-bit counts, sparse lists (a dense 4^13 histogram would be 512 MiB)
    n4, len4, k4 = 8000, 20000, 13
    codes, _ = synth.family_codes(20260004, n4, len4)
    s4 = api.HistogramSet(ctx, k4, 64, n4, sparse_entries=n4 * (len4 + 64))
    for done in range(0, n4, 2000):
        b = synth.pack_batch(codes[done:done + 2000])
        s4.build_packed(done, len(codes[done:done + 2000]), b["packed"], b["n_bases"], b["seg_seq"], b["seg_start"], b["seg_end"], b["eff_len"], b["one_mers"])
    w4 = wtext.replace("k: %d" % args.k, "k: %d" % k4).replace("uint%d_t" % args.dtype, "uint64_t")
    trn4 = api.Trainer(ctx, api.Feature.from_text(ctx, w4, 0), 0.9)
    ent4 = int(8 * np.mean([s4.entries(i) for i in range(0, n4, 16)]))
    rows.append(get_close_leg(api, ctx, trn4, s4, n4, 20, ent4, "cfg4's shape: %d x %d bp, k=%d, datatype=64, sparse layout" % (n4, len4, k4)))
    del s4
    # r05: the kernels that cost the clustering runs their wall time -- the `--feat slow` model, the window pass on mixed lengths, get_mean
    for leg in (lambda: slow_model_leg(api, ctx, hs, M, args), lambda: window_leg(api, synth, ctx, args), lambda: mean_nearest_leg(api, ctx, hs, M, args)):
        try:
            rows.append(leg())
        except Exception as e:      # noqa: BLE001 -- one leg failing must not cost the others
            rows.append({"workload": "failed", "error": repr(e)})
    return rows


def slow_model_leg(api, ctx, hs, M, args):
    """Q x M with BASELINE cfg5's model (`--feat slow`: 5 statistics incl. jensen_shannon, predict/Feature.cpp:984-1009) on the resident
    set: the integer reductions come from the matrix-core pass as in the main leg, the divergence sums from one merge pass per query over
    the sets' sparse mirrors queued behind it. Bytes: 8 per stored bin of a candidate's list, once per query."""
    wtext = open(os.path.join(ROOT, "tests", "golden", "weights_cfg5_k9.txt")).read().replace("uint8_t", "uint%d_t" % args.dtype)
    feat = api.Feature.from_text(ctx, wtext, 0)
    nq = 256
    out = {"close": api.pinned_array(ctx, (nq, M), np.uint8)}
    qs = np.array([(j * 7919 + 11) % M for j in range(nq)], dtype=np.uint32)
    api.score_multi(ctx, feat, hs, None, hs, qs[:128], m=M, want=("close", "counts"), out={"close": out["close"][:128]})          # (builds the mirrors: untimed)
    ctx.synchronize()
    t0 = time.perf_counter()
    res = api.score_multi(ctx, feat, hs, None, hs, qs, m=M, want=("close", "counts"), out=out)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    kernel, _ = ctx.last_kernel_info()
    ent = int(8 * np.mean([hs.entries(i) for i in range(0, M, max(1, M // 500))]))
    alg = nq * (M + 1) * ent
    ach = alg / dt / 1e9
    return {"workload": "the resident set, Q x M with the `--feat slow` model of cfg5 (weights_cfg5_k9.txt: %d statistics incl. jensen_shannon): %d queries x %d" % (feat.n_singles, nq, M),
            "metric": "sequence-pairs/sec identity-scored, Q x M, divergence statistics included", "value": nq * M / dt, "unit": "pairs/s", "ms_per_128_queries": dt / nq * 128 * 1e3,
            "close_pairs": int(np.sum(res["counts"])),
            "roofline": {"bound": "hbm", "kernel": "k_pair_sparse_mp (one divergence pass per query over the mirrors' lists) behind " + kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes": alg, "seconds": dt,
                         "note": "wall clock of the whole call: product, rank walk, %d merge passes, epilogue" % nq}}


def window_leg(api, synth, ctx, args):
    """Trainer::get_close over the accumulate loop's window (cluster/ClusterFactory.cpp:553-610 -> cluster/Trainer.cpp:23-71) on a
    cfg5-shaped set: 8 000 sequences of 500 b - 50 kb (log-uniform templates, families of 5 at graded divergence), k = 9, 16-bit,
    sparse layout, the `--feat slow` model at --id 0.6: the (candidate, round) pass over rank lists (k_pair_ranks_items). Bytes: 4 per
    k-mer of every candidate inside the pass's length window."""
    seed, n_t, per = 20260005, 1600, 5          # (r05: 8 000 sequences -- at 2 000 a pass held a few hundred candidates and timed its launches, not the walk)
    seqs = []
    for t in range(n_t):
        u = synth._unit(synth._stream(seed, 2 * t + 1, 2))
        length = int(round(np.exp(np.log(500.0) + float(u[0]) * (np.log(50000.0) - np.log(500.0)))))
        tmpl = synth.template(seed, t, length)
        for j in range(per):
            rate = 0.03 + 0.07 * j
            seqs.append(synth.to_ascii(synth.member(seed, t, j, tmpl, sub_rate=rate * 0.85, indel_rate=rate * 0.15)))
    n = len(seqs)
    lens = np.array([len(s_) for s_ in seqs])
    hs = api.HistogramSet(ctx, 9, 16, n, sparse_entries=int(lens.sum()) + 4096)
    for off in range(0, n, 256):
        hs.build(seqs[off:off + 256], first_slot=off)
    wtext = open(os.path.join(ROOT, "tests", "golden", "weights_cfg5_k9.txt")).read().replace("uint8_t", "uint16_t")
    trn = api.Trainer(ctx, api.Feature.from_text(ctx, wtext, 0), 0.6)
    order = np.argsort(lens, kind="stable").astype(np.uint32)
    sl = lens[order]
    win = api.Window(ctx, hs, order)
    rng = np.random.default_rng(5)
    queries = [int(x) for x in rng.permutation(n)[:70]]

    def one(q):
        a, b = int(np.searchsorted(sl, int(lens[q] * 0.6), "left")), int(np.searchsorted(sl, int(lens[q] / 0.6), "right"))
        alive = win.alive(a, b)
        closed = win.get_close(trn, a, b, hs, q)[0]
        return alive, int(sl[a:b].sum()), len(closed), ctx.last_kernel_ms()[0]
    for q in queries[:10]:
        one(q)
    ctx.synchronize()
    pairs = kmers = 0
    kms = []
    kernels = {}
    t0 = time.perf_counter()
    for q in queries[10:]:
        alive, km, _, ms = one(q)
        pairs += alive
        kmers += km
        kms.append(ms)
        kernels[ctx.last_kernel_info()[0]] = kernels.get(ctx.last_kernel_info()[0], 0) + 1
    ctx.synchronize()
    dt = time.perf_counter() - t0
    alg = 4 * kmers          # (an upper bound: the k-mers of every sequence inside the length window, alive or not)
    k_s = float(np.sum(kms)) * 1e-3
    ach = alg / k_s / 1e9 if k_s > 0 else float("nan")
    return {"workload": "cfg5's shape: %d sequences of 500 b - 50 kb, k=9, datatype=16, sparse layout, `--feat slow` model, --id 0.6; %d window passes of msc_get_close_window" % (n, len(kms)),
            "metric": "sequence-pairs/sec identity-scored inside the length windows, 1 x M window passes", "value": pairs / dt, "unit": "pairs/s", "passes": len(kms),
            "candidates_per_pass": pairs / max(1, len(kms)), "ms_per_pass": dt / max(1, len(kms)) * 1e3,
            "roofline": {"bound": "hbm", "kernel": max(kernels, key=kernels.get), "kernels": kernels, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                         "avg_launch_ms": float(np.mean(kms)), "algorithmic_bytes_per_launch": alg / max(1, len(kms)), "launches_timed": len(kms),
                         "note": "rank-list bytes (4 per k-mer) over the streaming kernel's own time (HIP events); value uses the wall clock of the passes"}}


def mean_nearest_leg(api, ctx, hs, M, args):
    """get_mean + closest (cluster/ClusterFactory.cpp:338-380, cluster/Trainer.cpp:144-157) on 1 000 members of the resident dense set:
    column sums -> FP64 mean -> distance_d of every member to it -> first minimum. SURVEY 8(d): 2 m 4^k sizeof(T) bytes per call."""
    m = min(1000, M)
    slots = np.array([(j * 97 + 5) % M for j in range(m)], dtype=np.uint32)
    api.mean_nearest(ctx, hs, slots)
    ctx.synchronize()
    calls = 5
    t0 = time.perf_counter()
    for _ in range(calls):
        pos, d, _ = api.mean_nearest(ctx, hs, slots)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / calls
    alg = 2 * m * (4 ** args.k) * (args.dtype // 8)
    ach = alg / dt / 1e9
    return {"workload": "get_mean + closest over %d members of the resident set (k=%d, datatype=%d, dense)" % (m, args.k, args.dtype), "metric": "mean_and_nearest calls/s", "value": 1.0 / dt,
            "unit": "calls/s", "ms_per_call": dt * 1e3, "members": m, "nearest_position": int(pos),
            "roofline": {"bound": "hbm", "kernel": "k_colsum + k_pair_tiles + k_distance_members", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": alg, "note": "wall clock of the call (three kernels + the copy of the distances home)"}}


def one_stream_product(api, ctx, feat, hs, M, Q, n_total):
    """The product kernel's roofline without a neighbour: two steps with msc_set_block_pipe(0) (every kernel of a block on one stream), the
    product timed by the library's HIP events on that stream."""
    ctx.set_block_pipe(False)
    try:
        nq = min(Q, 1024)
        out = {"close": api.pinned_array(ctx, (nq, M), np.uint8)}
        ms, launches = [], []
        for st in range(3):
            qs = np.array([((st * nq + j) * 7919 + 3) % M for j in range(nq)], dtype=np.uint32)
            api.score_multi(ctx, feat, hs, None, hs, qs, m=M, want=("close", "counts"), out=out)
            if st:
                ms.append(ctx.last_kernel_ms()[0])
                launches.append(ctx.last_kernel_launches())
        avg = float(np.sum(ms)) / max(1, int(np.sum(launches)))
        return avg, int(np.sum(launches))
    finally:
        ctx.set_block_pipe(True)


def pmc_traffic(kernel_prefix, config_key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
    (profiles/*_pmc_hbm.json, written by tools/summarize_profiles.py: 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction), and the
    VALU-busy fraction of the same passes where they were taken. (None, None) when no committed profile matches the running configuration."""
    import glob
    best, busy = None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm.json"))):
        try:
            d = json.load(open(f))
        except Exception:      # noqa: BLE001
            continue
        if d.get("_config") != config_key:
            continue
        for k, v in d.items():
            if k.startswith(kernel_prefix) and isinstance(v, dict) and "hbm_bytes_per_launch_corrected" in v:
                best = v["hbm_bytes_per_launch_corrected"]
                busy = v.get("valu_busy")
    return best, busy


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        # MSC_BENCH_BACKEND=gloo + MSC_BENCH_ONE_GPU=1 lets two ranks share one device to smoke-test the exchange logic on a
        # 1-GPU box; the real launch (driver) is one rank per GPU over RCCL ("nccl" IS RCCL on ROCm).
        backend = os.environ.get("MSC_BENCH_BACKEND", "nccl")
        if os.environ.get("MSC_BENCH_ONE_GPU") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        real = dist

        class _HostStaged:
            """every collective of the exchange through host memory over a gloo group: the 1-GPU smoke test (two ranks share a device),
            and the fallback if RCCL cannot move the library's device views on this node (preflight below)"""
            def __init__(self, group):
                self.group = group

            def __getattr__(self, name):
                return getattr(real, name)

            class _Done:
                def wait(self):
                    return True

            def all_gather_into_tensor(self, out, inp, async_op=False):
                ho, hi = out.cpu(), inp.cpu().contiguous()
                real.all_gather_into_tensor(ho, hi, group=self.group)
                out.copy_(ho)
                return self._Done()

            def broadcast(self, t, src, async_op=False):
                h = t.cpu()
                real.broadcast(h, src=src, group=self.group)
                t.copy_(h)
                return self._Done()

            def all_gather(self, outs, t):
                hs_ = [o.cpu() for o in outs]
                real.all_gather(hs_, t.cpu(), group=self.group)
                for o, h in zip(outs, hs_):
                    o.copy_(h)

            def all_reduce(self, t, op=None):
                h = t.cpu()
                real.all_reduce(h, op=op if op is not None else real.ReduceOp.SUM, group=self.group)
                t.copy_(h)

            def barrier(self):
                real.barrier(group=self.group)

        if backend == "nccl":
            import datetime
            real.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=600))
            gloo_group = real.new_group(backend="gloo", timeout=datetime.timedelta(seconds=180))
        else:
            real.init_process_group(backend)
            gloo_group = None
            dist = _HostStaged(None)
    from meshclust2_amd import api, shard, synth

    n_total = args.nseq if args.scaling == "strong" else args.nseq * world
    plan = shard.ShardPlan(n_total, world, block=BLOCK)
    Q = args.queries
    if world > 1 and args.layout == "sparse":
        raise SystemExit("--layout sparse is a single-GPU comparison (the exchange of 8(e) moves dense slots)")
    if Q % world:
        raise SystemExit("--queries must be a multiple of the number of ranks (every rank contributes queries / N per step)")
    ctx = api.Context(local_rank)
    by_rows = args.shard == "rows" and args.mode == "allpairs" and args.layout == "dense"
    rows_job = None
    if by_rows and world > 1:
        # every rank generates its share of the sequences on the host, then all of them build ALL candidates (slot = global index)
        _, gen_s, _, (seq_rows, one_rows) = build_resident_set(api, synth, ctx, args, plan, rank, build=False)
        hs, build_s, build_dev_s, rows_job = build_replicated_set(api, shard, torch, dist, ctx, args, plan, rank, seq_rows, one_rows)
        build_s += gen_s
        M = n_total
    else:
        hs, build_s, build_dev_s, (seq_rows, one_rows) = build_resident_set(api, synth, ctx, args, plan, rank, tail=0 if world == 1 else None)
        M = plan.local_count(rank)                              # this rank's shard
    m_min = min(plan.local_count(r) for r in range(world))
    wpath = args.weights or os.path.join(ROOT, "tests", "golden", "weights_k9_u32.txt" if args.k == 9 else "weights_k5_u16.txt")
    wtext = open(wpath).read()
    feat = api.Feature.from_text(ctx, wtext, 0)
    trn = api.Trainer(ctx, feat, 0.9)

    # N > 1: the exchange of SURVEY 8(e) through meshclust2_amd/shard.py; the collectives land directly in slots M .. M + 2Q - 1 (two
    # blocks: the next step's queries arrive while this step's are scored)
    sharded = block = None
    if world > 1 and rows_job is None:
        all_bins, all_scal = shard.device_tensors(hs, M + 2 * Q)      # [slot, bytes] views of the set's device memory
        ship_seq = args.exchange == "sequences" and args.mode == "allpairs"
        if ship_seq:
            # What travels per step is the bases only. The lengths and 1-mer counts of EVERY sequence are exchanged once here (40 bytes
            # each): the host then knows the segment table of any query block without reading device memory, and the block's histograms
            # are built straight from the gathered device buffer (msc_hist_build_packed_dev) -- no D2H / H2D hop inside a step.
            seq_dev = torch.from_numpy(seq_rows).cuda()
            qseq_dev = torch.zeros((2 * Q, seq_rows.shape[1]), dtype=torch.uint8, device="cuda")
            m_max = max(plan.local_count(r) for r in range(world))
            mine = torch.zeros((m_max, 5), dtype=torch.int64)
            mine[:M] = torch.from_numpy(one_rows.view(np.int64))
            every = [torch.zeros_like(mine) for _ in range(world)]
            (dist if gloo_group is None else real).all_gather(every, mine, **({} if gloo_group is None else {"group": gloo_group}))
            meta_of_rank = [t.numpy().view(np.uint64) for t in every]          # [rank][local slot] = (1-mer counts x 4, length)

        block_out = {}
        first_of_base = {}          # query buffer half -> first local slot of the block that was gathered into it (every rank contributes first .. first + Q / N - 1)

        class GpuBackend:
            def query_buffers(self, j=0):
                return [all_bins[M + j], all_scal[M + j]]

            def export_query(self, local):
                return [all_bins[local], all_scal[local]]

            def export_block(self, local_first, n):
                if ship_seq:
                    return [seq_dev[local_first:local_first + n]]
                return [all_bins[local_first:local_first + n], all_scal[local_first:local_first + n]]

            def block_buffers(self, base, n_rows):
                if ship_seq:
                    return [qseq_dev[base:base + n_rows]]
                return [all_bins[M + base:M + base + n_rows], all_scal[M + base:M + base + n_rows]]


            def import_query(self):
                self.import_queries(1)

            def import_queries(self, n, base=0):
                torch.cuda.synchronize()          # the all-gather (issued a step ago on RCCL's stream) has landed; the library has its own stream
                if ship_seq and n > 1:
                    # the block arrived as sequences: its histograms are built here, from the device buffer the collective filled (one
                    # segment per sequence, every sequence on its own bytes); lengths and 1-mer counts from the tables exchanged at setup
                    first, per = first_of_base[base], n // world
                    meta = np.concatenate([meta_of_rank[r][first:first + per] for r in range(world)])
                    starts = np.arange(n, dtype=np.uint64) * np.uint64(seq_rows.shape[1] * 4)
                    lens = meta[:, 4].copy()
                    hs.build_packed_dev(M + base, n, qseq_dev[base:base + n].data_ptr(), n * seq_rows.shape[1] * 4, np.arange(n, dtype=np.uint32), starts,
                                        starts + lens - np.uint64(1), lens, np.ascontiguousarray(meta[:, :4]).reshape(-1))
                    return
                hs.import_done(M + base, n)

            def score_local(self):
                flags, bp, bs, _ = trn.get_close(hs, None, hs, M, m=M)
                return flags, bp, bs

            def score_block(self, n, base=0):
                if n not in block_out:          # the block's product (one close flag per pair) lands in page-locked memory allocated once
                    block_out[n] = {"close": api.pinned_array(ctx, (n, M), np.uint8)}
                res = api.score_multi(ctx, feat, hs, None, hs, np.arange(M + base, M + base + n, dtype=np.uint32), m=M, want=("close", "counts"), out=block_out[n])
                return res["close"], res["counts"]

        if gloo_group is not None:
            # preflight: one all-gather and one broadcast over RCCL on the library's own device views (they are not torch allocations).
            # If any rank fails, every rank stages the exchange through host memory instead of aborting the run.
            # The collectives are issued asynchronously and waited for with a deadline: a rank that throws at once does not leave the
            # others blocked inside the collective for the process group's whole timeout -- they time out here, everybody meets in
            # the flag all-reduce below (gloo, its own deadline), and a TIMEOUT ends the run non-zero (the communicator is then in an
            # unknown state), while a refusal every rank sees immediately falls back to host staging.
            import datetime
            ok = 2
            try:
                w1 = real.all_gather_into_tensor(all_scal[M:M + world], all_scal[0:1], async_op=True)
                w2 = real.broadcast(all_scal[M], src=0, async_op=True)
                for w in (w1, w2):
                    w.wait(timeout=datetime.timedelta(seconds=60))
                torch.cuda.synchronize()
            except Exception as e:      # noqa: BLE001
                timed_out = "imeout" in repr(e) or "imed out" in repr(e)
                sys.stderr.write("rank %d: RCCL preflight on device views %s (%r)\n" % (rank, "TIMED OUT" if timed_out else "failed: staging the exchange through host memory", e))
                ok = 0 if timed_out else 1
            flag = torch.tensor([ok], dtype=torch.int32)
            real.all_reduce(flag, op=real.ReduceOp.MIN, group=gloo_group)
            if int(flag.item()) == 0:
                sys.stderr.write("rank %d: a rank timed out in the RCCL preflight: giving up\n" % rank)
                os._exit(3)
            if int(flag.item()) == 1:
                dist = _HostStaged(gloo_group)
        sharded = shard.ShardedTrainer(dist, plan, GpuBackend(), rank, device="cuda")
        block = shard.ShardedBlockScorer(dist, plan, GpuBackend(), rank, device="cuda")

    tiles_ms, launches = [], []
    totals = []                     # N > 1: the close counts' all-gathers in flight (shard.ShardedBlockScorer.score(defer=True))
    checks = []                     # --check: per timed step, close candidates per query of the block (all ranks)
    cw = args.check_world if world == 1 and args.check_world > 1 else 0
    if cw:
        plan_w = shard.ShardPlan(n_total, cw, block=BLOCK)
        m_min_w = min(plan_w.local_count(r) for r in range(cw))
    # the step's product -- one close flag per pair, Q x M bytes -- lands in page-locked host memory allocated once
    qpr = Q // world                # queries every rank contributes per step (--shard candidates) / scores per step (--shard rows)
    out_close = {"close": api.pinned_array(ctx, (qpr if rows_job else Q, M), np.uint8)} if args.mode == "allpairs" and (world == 1 or rows_job) else None
    step_no = [0]
    pending = [None, None]          # RCCL work handles of the query block in flight per buffer half (N > 1, allpairs)
    if rows_job is not None:
        def score_rows(globals_):
            res = api.score_multi(ctx, feat, hs, None, hs, np.asarray(globals_, dtype=np.uint32), m=M, want=("close", "counts"), out=out_close)
            return res["counts"]
        rows_job.backend.score_rows = score_rows

    def block_first(st):
        """first local slot of the queries / N consecutive histograms every rank contributes to the block of step st"""
        return (st * qpr * 131) % (m_min - qpr + 1)

    def one_step():
        """`queries` query histograms against every resident histogram of every rank"""
        st = step_no[0]
        step_no[0] += 1
        if args.mode == "allpairs":
            if rows_job is not None:
                # --shard rows: this rank's queries / N rows of the step's query list (the list one rank would score) against ALL
                # candidates; the counts' all-gather is issued here and read one step later
                qs = [((st * Q + j) * 7919) % n_total for j in range(Q)]
                totals.append(rows_job.score(qs, defer=True))
                if len(totals) > 1:
                    done = totals.pop(0).total()
                    if args.check:
                        checks.append([int(x) for x in done])
            elif block is not None:
                # double-buffered exchange: this block's queries were issued during the previous step (or are issued now, on the
                # first one); the NEXT block's all-gathers go out before this block is scored and run underneath it
                cur, nxt = st % 2, (st + 1) % 2
                if pending[cur] is None:
                    first_of_base[cur * Q] = block_first(st)
                    pending[cur] = block.begin_packed(block_first(st), qpr, base=cur * Q)
                block.finish(pending[cur], Q, base=cur * Q)
                pending[cur] = None
                first_of_base[nxt * Q] = block_first(st + 1)
                pending[nxt] = block.begin_packed(block_first(st + 1), qpr, base=nxt * Q)
                # the per-query close counts of the block (all ranks summed) are the step's result; their all-gather is issued here and
                # waited for one step later (or at the end): nothing in this step needs them
                _, total = block.score(Q, base=cur * Q, defer=True)
                totals.append(total)
                if len(totals) > 1:
                    done = totals.pop(0).total()
                    if args.check:
                        checks.append([int(x) for x in done])
            else:
                if cw:          # the block a cw-rank run assembles at this step: rank r's local slots first .. first + Q / cw - 1, rank-major
                    first = (st * (Q // cw) * 131) % (m_min_w - Q // cw + 1)
                    qs = np.array([plan_w.global_index(r, first + j) for r in range(cw) for j in range(Q // cw)], dtype=np.uint32)
                else:
                    qs = np.array([((st * Q + j) * 7919) % M for j in range(Q)], dtype=np.uint32)
                res = api.score_multi(ctx, feat, hs, None, hs, qs, m=M, want=("close", "counts"), out=out_close)
                if args.check:
                    checks.append([int(x) for x in res["counts"]])
            tiles_ms.append(ctx.last_kernel_ms()[0])
            launches.append(ctx.last_kernel_launches())
        else:
            for j in range(Q):
                g = ((st * Q + j) * 7919) % n_total
                if sharded is not None:
                    sharded.get_close(g)
                else:
                    trn.get_close(hs, None, hs, g, m=M)
                tiles_ms.append(ctx.last_kernel_ms()[0])
                launches.append(ctx.last_kernel_launches())

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        one_step()
    for tt in totals:
        tt.total()
    totals.clear()
    tiles_ms.clear()
    launches.clear()
    checks.clear()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    for half in pending:             # the block issued by the last timed step: every rank issued it, so it completes
        for w in half or []:
            w.wait()
    for tt in totals:                # the counts of the last step
        done = tt.total()
        if args.check:
            checks.append([int(x) for x in done])
    totals.clear()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    pairs = args.steps * Q * n_total
    esz = args.dtype // 8
    # 1 x M streaming = 4^k*sizeof(T) bytes per scored pair. Q x M: the library reports which kernel ran and how
    # many queries ONE HBM read of a candidate tile served in it (16 for the digest kernel: one workgroup scores 16 queries
    # per tile it fetches) -> algorithmic bytes of a launch = candidates x histogram bytes x ceil(Q / that) + the query tiles.
    # Per-rank figures (rank 0's shard: M candidates).
    kernel, qtile = ctx.last_kernel_info()
    hist_bytes = (4 ** args.k) * esz
    on_mfma = on_fp4 = kernel.startswith("k_pair_gemm_fp4_dma")
    qblk = 128          # queries per pass over the candidates on the matrix cores (the library's block)
    if on_mfma:   # everything from the matrix cores: the pass reads the presence-bit mirror, one BIT per bin, once per block of up to 128 queries
        hist_bytes = 4 ** args.k // 8
    elif "no emd" in kernel or "emd by ranks" in kernel:   # count-only form of the digest kernel: the prefix half of each tile is not fetched
        hist_bytes //= 2
    if args.layout == "sparse":            # a pair reads the candidate's entry list: 8 bytes per stored bin (mean over a sample of slots)
        hist_bytes = int(8 * np.mean([hs.entries(i) for i in range(0, M, max(1, M // 500))]))
    if kernel.startswith("k_pair_ranks"):          # a pass over rank lists reads 4 bytes per k-mer of a candidate (secondary_legs counts the same)
        hist_bytes = 4 * (args.length - args.k + 1)
    n_launch = max(int(np.sum(launches)), 1)
    avg_ms = float(np.sum(tiles_ms)) / n_launch if tiles_ms else float("nan")
    q_call = qpr if rows_job is not None else Q          # query rows one library call of this rank scores
    if args.mode == "allpairs" and args.layout == "sparse":
        per_call = Q * (M + 1) * hist_bytes          # one 1 x M merge pass per query
    elif args.mode == "allpairs":
        per_call = (M * -(-q_call // qtile) + q_call) * hist_bytes      # (qtile <= 128: a call with more queries is that many passes)
    else:
        per_call = (M + 1) * hist_bytes
    # one timed call may be several launches of the streaming kernel (candidate chunks): report per launch
    alg_bytes = per_call * len(tiles_ms) // n_launch
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    config_key = "nseq=%d,len=%d,k=%d,dtype=%d,queries=%d,mode=%s,kernel=%s" % (M, args.length, args.k, args.dtype, Q, args.mode, kernel)
    if args.repeats:
        config_key += ",repeats=%g" % args.repeats
    if args.weights:
        config_key += ",weights=" + os.path.basename(args.weights)
    if args.layout == "sparse":
        config_key += ",layout=sparse"
    traffic, valu_busy = pmc_traffic(kernel.split("<")[0], config_key)
    per_gpu = "%d per GPU" % args.nseq if args.scaling == "weak" and rows_job is None else "%d in total (%d on rank 0)" % (n_total, M)
    length_name = "%gkb" % (args.length / 1000.0) if args.length % 100 == 0 else "%d bp" % args.length
    is_cfg2 = (n_total, args.length, args.k, args.dtype, args.layout) == (100000, 1000, 9, 32, "dense")
    calls_per_launch = len(tiles_ms) / n_launch if tiles_ms else 1.0          # < 1: a call is cut into candidate chunks, one launch each
    line = {
        "metric": "sequence-pairs/sec identity-scored (k=%d, %s seqs)" % (args.k, length_name),
        "value": pairs / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "u%d" % args.dtype, "data": "synthetic",
        "config": {"workload": "%s: %d x %d bp synthetic sequences, %s, k=%d, datatype=%d, %s layout, resident in HBM; step = %d query histograms x all "
                               "histograms (model %s: %d single statistics + %d-combo GLM + close flag per pair), mode %s"
                               % (("cfg2" if not args.repeats else "cfg2 with a 400-base repeat in %g %% of the sequences" % (100 * args.repeats)) if is_cfg2 else "custom (not a BASELINE config)", n_total, args.length, per_gpu, args.k, args.dtype, args.layout, Q, os.path.basename(wpath),
                                  feat.n_singles, feat.n_combos, args.mode),
                   "pairs_per_step": Q * n_total, "pairs_timed": pairs,
                   "projected_seconds_full_matrix": round(float(n_total) ** 2 / (pairs / dt), 1), "hist_bytes": hist_bytes if args.layout == "sparse" else (4 ** args.k) * esz,
                   "streamed_bytes_per_candidate_read": hist_bytes,
                   "layout": args.layout, "hbm_resident_gib": hs.nbytes() / 2 ** 30,
                   "build_seconds": round(build_s, 2),
                   "hist_build": {"seq_per_s": round(build_dev_s["median_chunk"]), "GBps_written": round(build_dev_s["median_chunk"] * (4 ** args.k) * esz / 1e9, 1),
                                  "seq_per_s_all_chunks": round(build_dev_s["all"]),
                                  "note": "msc_hist_build_packed only (2-bit bases over PCIe + build kernel): the last chunk rebuilt in place back to "
                                          "back; all_chunks = the chunks as first built, each after ~1 s of host-side sequence generation on an idle GPU; "
                                          "untimed setup"},
                   "sharding": ("query rows: every rank holds all %d candidates (set-up: one RCCL all-gather of the 2-bit sequences + one of their 1-mer / length records, "
                                "each rank builds every histogram) and scores %d of each step's %d query rows against all of them; no collective on the data path, "
                                "the per-query close counts are all-gathered one step late" % (n_total, qpr, Q)) if rows_job is not None else
                               ("sequence blocks of %d, block-cyclic; per step 2 all-gathers assemble the query block (RCCL) as %s, per-query close counts all-gathered"
                                % (BLOCK, "2-bit packed sequences + 1-mer counts (every rank builds the block's histograms)" if args.exchange == "sequences" else "histograms"))
                               if world > 1 else "single GPU",
                   "ms_per_1024_queries": dt / args.steps * 1e3 * 1024.0 / Q,
                   "queries_per_candidate_read": qtile},
        "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": alg_bytes, "launches_timed": n_launch,
                     "algorithmic_GBps": achieved,
                     # what the hardware did, from the committed counter passes of this command (null without a matching profile):
                     # hbm_frac = PMC HBM bytes per launch / launch time / peak (frac above prices the TILE reads of the kernel as it
                     # is tiled -- query groups that share a candidate mostly hit in L2); valu_busy = VALU cycles / busy cycles
                     "hbm_frac": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and avg_ms == avg_ms else None, "valu_busy": valu_busy,
                     # a call scores its queries in blocks of 128 (matrix cores) or 64 (one pass over the candidates each) and may cut a pass into candidate chunks
                     "candidates_per_launch": int(round(M * calls_per_launch * (-(-q_call // (qblk if on_mfma else 64)) if args.mode == "allpairs" and args.layout == "dense" else 1))),
                     "query_groups_per_launch": -(-min(q_call, qblk if on_mfma else 64) // qtile) if args.mode == "allpairs" else 1,
                     "profile_key": config_key},
    }
    if on_mfma and args.mode == "allpairs" and avg_ms == avg_ms:
        # The pass on the matrix cores is bound by the matrix pipe (FP4 operands), not by HBM (it streams one bit per bin): 4^k multiply-adds per
        # pair = 2 * 4^k integer operations; a launch scores (candidates of the launch) x (rows of its query block, padded rows included
        # in the work the pipe does but NOT in the operations counted here).
        pairs_per_launch = float(M) * q_call * len(tiles_ms) / n_launch
        ops = 2.0 * pairs_per_launch * (4 ** args.k)
        tops = ops / (avg_ms * 1e-3) / 1e12
        peak = MFMA_FP4_PEAK_TOPS if on_fp4 else MFMA_I8_PEAK_TOPS
        line["roofline"].update({"bound": "mfma", "achieved": tops, "peak": peak, "unit": "TFLOP/s", "frac": tops / peak,
                                 "ops": ("FP4 (E2M1) multiply-adds counted as 2 operations each (v_mfma_f32_32x32x64_f8f6f4, operands 0 / 0.5 / 1 / 2: every "
                                         "product of two set presence bits is exactly 1.0), exact sums in f32 (< 2^24)") if on_fp4 else
                                        "int8 multiply-adds counted as 2 operations each (v_mfma_i32_32x32x32_i8), exact int32 sums",
                                 "algorithmic_ops_per_launch": ops})
    if on_mfma and args.mode == "allpairs" and world == 1 and not args.check:
        # the same kernel without a neighbour on the chip: in the timed region above three streams share the CUs, and a kernel's launch time
        # there includes what the others took from it
        try:
            one_ms, one_n = one_stream_product(api, ctx, feat, hs, M, Q, n_total)
            one_tops = 2.0 * float(M) * 128 * (4 ** args.k) / (one_ms * 1e-3) / 1e12
            line["roofline"]["one_stream"] = {"avg_launch_ms": one_ms, "launches_timed": one_n, "achieved": one_tops, "peak": MFMA_FP4_PEAK_TOPS, "unit": "TFLOP/s", "frac": one_tops / MFMA_FP4_PEAK_TOPS,
                                              "note": "msc_set_block_pipe(0): every kernel of a block on one stream; the figure profiles/*_kernel_stats.csv of the same round holds"}
        except Exception as e:      # noqa: BLE001
            line["roofline"]["one_stream"] = {"error": repr(e)}
    if traffic is not None:
        line["roofline"]["traffic_source"] = "profiles/*_pmc_hbm.json: the committed rocprofv3 --pmc passes of this command (not measured in this run)"
    if args.check:
        line["check"] = checks
    if world == 1 and args.secondary and args.mode == "allpairs" and args.layout == "dense":
        try:
            line["secondary"] = secondary_legs(api, synth, ctx, args, hs, M, seq_rows, one_rows, wtext)
        except Exception as e:      # noqa: BLE001 -- the main line must not depend on the extra legs
            line["secondary"] = [{"workload": "failed", "error": repr(e)}]
    if rank == 0:
        if args.cpu_seconds > 0 and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline(args, synth, wtext, wpath)
            except Exception as e:      # noqa: BLE001
                line["cpu_baseline"] = {"value": None, "unit": "pairs/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
