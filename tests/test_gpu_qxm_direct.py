"""The Q x M pass (msc_score_multi, fastcar's work() shape, fastcar/FC_Runner.cpp:426-471) held FIRST-HAND to the reference: its raw
statistics, weighted sums, classification values and close flags next to the reference-generated fixtures (tests/golden/vectors_*.npz)
and next to the CPU oracle (predict/Feature.cpp:156-171 compute_all_raw -> normalize_cache -> operator(); predict/Predictor.cpp:
315-333 classify_sum / p_close; cluster/Trainer.cpp:49-52) -- not through another HIP kernel of the same library. The kernel the
library picked is asserted by name wherever the route rule says the matrix-core pass must run."""
import os
import sys

import numpy as np
import pytest

from golden_util import EXACT, FEATS, VECTOR_SETS, load_vectors, weights_text
from meshclust2_amd import api, synth

pytestmark = pytest.mark.gpu
ALL_MASK = sum(1 << b for _, b in FEATS)
FAST_MASK = sum(1 << b for name, b in FEATS if name not in ("jefferey_divergence", "jensen_shannon"))
GEMM_KERNELS = ("k_pair_gemm_fp4_dma<",)         # the pass on the matrix cores, as msc_last_kernel_info names it: "k_pair_gemm_fp4_dma<N query rows, ...>", the one product kernel
RTOL = 1e-9


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def _held_to(raw, exp, where):
    """raw, exp: [..., 11] in FEATS order; integer-derived statistics bitwise, the FP64 sums to 1e-9"""
    for col, (name, _) in enumerate(FEATS):
        if name in EXACT and name != "kulczynski2":
            assert np.array_equal(raw[..., col], exp[..., col]), (name, where)
        else:
            assert np.allclose(raw[..., col], exp[..., col], rtol=RTOL, atol=1e-13), (name, where)


@pytest.mark.parametrize("vec,wts", VECTOR_SETS)
def test_golden_vectors_through_the_qxm_pass(ctx, vec, wts):
    """Every query of a fixture set against every histogram of it in ONE msc_score_multi call, next to the reference's own numbers."""
    v = load_vectors(vec)
    k, dt, n = int(v["k"]), int(v["dtype"]), int(v["n"])
    hs = api.HistogramSet(ctx, k, dt, n)
    hs.build([bytes(s) for s in v["seqs"]])
    feat = api.Feature.from_text(ctx, weights_text(wts), 0)
    everyone = np.arange(n, dtype=np.uint32)
    for order in (api.ORDER_CAND_FIRST, api.ORDER_QUERY_FIRST):
        got = api.score_multi(ctx, feat, hs, everyone, hs, everyone, order=order, feat_mask=ALL_MASK, want=("sum", "csum", "close", "counts"))
        kernel = ctx.last_kernel_info()[0]
        if vec == "vectors_k9_u32.npz":          # dense 32-bit set, 4^9 bins, counts <= 3, 1 kb lists: the route rule says matrix cores
            assert kernel.startswith(GEMM_KERNELS), kernel
        # fixture layout: raw[first point, second point, statistic]; the pass returns [query][candidate]
        exp = np.transpose(v["raw"], (1, 0, 2)) if order == api.ORDER_CAND_FIRST else v["raw"]
        _held_to(got["raw"], exp, (vec, order, kernel))
        if order == api.ORDER_CAND_FIRST:          # the model values the reference printed are for (candidate, query)
            assert np.allclose(got["sum"], v["sums"].T, rtol=1e-8, atol=1e-10), (vec, kernel)
            assert np.allclose(got["csum"], v["csums"].T, rtol=RTOL), (vec, kernel)
            assert np.array_equal(got["close"], v["close"].T.astype(np.uint8)), (vec, kernel)
            assert np.array_equal(got["counts"], v["close"].T.astype(np.uint64).sum(axis=1)), (vec, kernel)


def _repeat_bearing(seqs, every, kind):
    """the construction of tests/ring_variant_check.py: a homopolymer / dinucleotide / 12-mer run spliced into every `every`-th sequence"""
    out = []
    for i, s in enumerate(seqs):
        s = bytes(s)
        if i % every == 1:
            at = 100 + 13 * (i % 50)
            run = {"homo": b"A" * 400, "di": b"AC" * 200, "unit12": b"ACGTTGCAAGTC" * 11, "unit3": b"ACG" * 40}[kind]
            s = s[:at] + run + s[at:]
        out.append(s)
    return out


@pytest.mark.parametrize("dtype,k,n,length,nq,repeats", [
    (32, 9, 150, 1000, 128, None),            # cfg2's shape: the matrix-core route, a whole block of queries
    (32, 9, 150, 1000, 37, "homo"),           # one count of ~390 in some sequences: what real FASTA looks like (VERDICT r03 missing #1)
    (32, 9, 150, 1000, 130, "di"),            # two counts of ~196, more queries than one block
    (16, 9, 90, 1000, 64, "unit12"),          # counts 9 .. 16
    (8, 9, 90, 1000, 20, "homo"),             # saturating 8-bit bins (255) among small counts
    (16, 8, 80, 2000, 9, "unit3"),            # counts ~40
    (32, 7, 120, 600, 70, None),              # k = 7: counts of 1 .. 5 in most bins' neighbourhood
    (8, 6, 60, 200, 5, None),                 # 4096 bins
    (32, 5, 60, 100, 33, "unit3"),            # 1024 bins of 32 bits: the smallest histogram the route takes
    (16, 5, 60, 150, 6, None),                # 2 KiB histograms: below the route (another kernel, same answers)
])
def test_qxm_pass_against_the_oracle(ctx, oracle, dtype, k, n, length, nq, repeats):
    """msc_score_multi next to oracle.raw_feature / oracle.score for sampled pairs of seeded sets, both argument orders, slot lists
    with repeats, queries from the candidates' own set -- whatever route the library picks; the matrix-core route asserted where its
    rule holds (dense, 8/16/32-bit, 4^k a multiple of 1024, lists at most a quarter of the bins)."""
    seqs, _ = synth.families(4100 + 31 * k + dtype + nq, n, length, family=6, length_jitter=length // 10)
    seqs = _repeat_bearing(seqs, 7, repeats) if repeats else [bytes(s) for s in seqs]
    hs = api.HistogramSet(ctx, k, dtype, n)
    hs.build(seqs)
    text = weights_text("weights_k9_u32.txt").replace("k: 9", "k: %d" % k).replace("uint32_t", "uint%d_t" % dtype)
    feat = api.Feature.from_text(ctx, text, 0)
    pred = oracle.predictor(text)
    rng = np.random.default_rng(k * 1000 + nq)
    q_slots = rng.integers(0, n, nq).astype(np.uint32)
    q_slots[:3] = (1, 8, 0)          # a repeat-bearing query (slot 1, 8), a plain one
    cands = np.concatenate([np.arange(n, dtype=np.uint32), rng.integers(0, n, 9).astype(np.uint32)])
    oh = [oracle.hist(s, k, dtype) for s in seqs]
    for order in (api.ORDER_CAND_FIRST, api.ORDER_QUERY_FIRST):
        got = api.score_multi(ctx, feat, hs, cands, hs, q_slots, order=order, feat_mask=FAST_MASK, want=("sum", "csum", "close", "counts"))
        kernel = ctx.last_kernel_info()[0]
        longest = max(hs.info(i)["sum"] for i in range(n)) - 4 ** k          # k-mers of the longest list
        if 4 ** k % 1024 == 0 and 4 ** k * dtype // 8 >= 4096 and longest * 4 <= 4 ** k and dtype != 64:
            assert kernel.startswith(GEMM_KERNELS), (kernel, dtype, k, repeats, longest)
        assert np.array_equal(got["counts"], got["close"].sum(axis=1, dtype=np.uint64))
        fast = [(name, b) for name, b in FEATS if (1 << b) & FAST_MASK]
        for qi in list(range(0, nq, max(1, nq // 6))) + [0, 1, 2]:
            q = int(q_slots[qi])
            for ci in list(range(0, len(cands), 11)) + [1, 8]:
                c = int(cands[ci])
                a, b = (oh[c], oh[q]) if order == api.ORDER_CAND_FIRST else (oh[q], oh[c])
                for col, (name, bit) in enumerate(fast):
                    exp = oracle.raw_feature(1 << bit, a, b)
                    val = got["raw"][qi][ci][col]
                    if name in EXACT and name != "kulczynski2":
                        assert val == exp, (name, q, c, order, kernel)
                    else:
                        assert val == pytest.approx(exp, rel=RTOL, abs=1e-13), (name, q, c, order, kernel)
                if order == api.ORDER_CAND_FIRST:
                    _, _, w = oracle.score(pred.cls, oh[c], oh[q])
                    assert got["sum"][qi][ci] == pytest.approx(w, rel=1e-8, abs=1e-10), (q, c, kernel)
                    assert got["close"][qi][ci] == (1 if round(1.0 / (1.0 + np.exp(-w))) > 0 else 0), (q, c, kernel)
    for h in oh:
        oracle.lib().orc_hist_free(h)


@pytest.mark.parametrize("seed", [11, 12, 15, 18, 21, 24, 27, 30])
def test_fuzz_gemm_route_seeds(ctx, seed):
    """Fixed seeds of tests/fuzz_gemm_route.py (the randomised checker aimed at the matrix-core route: k = 5 .. 9, three bin types, query
    blocks of 2 .. 200 from the same or another set, slot lists with repeats, both orders; every third seed splices short repeats in)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fuzz_gemm_route
    line = fuzz_gemm_route.one_round(ctx, seed)
    assert " ok:" in line, line


def test_fastcar_k9_u32_reproduces_reference_output_on_the_matrix_cores(tmp_path):
    """SURVEY 8(f4) at cfg2's histogram shape: the reference's own `fastcar --recover` output (tests/golden/fastcar_k9_u32.out,
    tests/golden/gen_golden.py make_fastcar_k9_output) for a 220 x 30 search at k = 9 / uint32_t whose database and queries include
    repeat-bearing sequences (counts up to ~290), reproduced byte for byte by msc_fastcar -- and the scoring passes ran on the
    matrix-core route (`--kernels` prints what msc_last_kernel_info named)."""
    import subprocess
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "meshclust2_amd", "host", "msc_fastcar")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "meshclust2_amd", "host")])
    db, h = synth.families(43, 220, 1000, family=10, length_jitter=120)
    q, hq = synth.families(43, 30, 1000, family=10, length_jitter=120)
    runs = [b"A" * 300, b"AC" * 150, b"ACGTTGCAAGTC" * 10]
    db = [s[:200 + i] + runs[(i // 9) % 3] + s[200 + i:] if i % 9 == 4 else s for i, s in enumerate(db)]
    q = [x[:len(x) - 5] for x in q]
    q[4] = q[4][:333] + runs[0] + q[4][333:]
    synth.write_fasta(str(tmp_path / "db.fa"), db, h)
    synth.write_fasta(str(tmp_path / "q.fa"), q, [x.replace(">seq", ">qry") for x in hq])
    golden = os.path.join(root, "tests", "golden")
    for extra in ([], ["--query-block", "30"]):
        r = subprocess.run([exe, "db.fa", "--query", "q.fa", "--recover", os.path.join(golden, "weights_k9_u32_fc.txt"), "--output", "fc_out", "--kernels"] + extra,
                           cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, (r.stdout + r.stderr).decode(errors="replace")[-2000:]
        got = open(str(tmp_path / "fc_out0"), "rb").read()
        exp = open(os.path.join(golden, "fastcar_k9_u32.out"), "rb").read()
        assert got == exp, "fastcar output differs (%d vs %d bytes)" % (len(got), len(exp))
        kernels = [ln.split(": ", 1)[1] for ln in r.stderr.decode().splitlines() if ln.startswith("kernel: ")]
        assert kernels and all(kn.startswith(GEMM_KERNELS) for kn in kernels), kernels


def test_fastcar_k5_u16_fixture_route(tmp_path):
    """Which kernel the k = 5 / 16-bit fastcar fixtures of tests/test_gpu_parity.py exercise: 2 KiB histograms are below the matrix-core
    route's rule (a slot under 4 KiB has no ranks mirror), so those byte-identical outputs pin the raw-bin Q x M kernels, not the GEMM."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "meshclust2_amd", "host", "msc_fastcar")
    db, h = synth.families(41, 300, 1000, family=10, length_jitter=150)
    q, hq = synth.families(41, 40, 1000, family=10, length_jitter=150)
    q = [x[:len(x) - 7] for x in q]
    synth.write_fasta(str(tmp_path / "db.fa"), db, h)
    synth.write_fasta(str(tmp_path / "q.fa"), q, [x.replace(">seq", ">qry") for x in hq])
    golden = os.path.join(root, "tests", "golden")
    r = subprocess.run([exe, "db.fa", "--query", "q.fa", "--recover", os.path.join(golden, "weights_k5_u16.txt"), "--output", "fc_out", "--kernels"],
                       cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr).decode(errors="replace")[-2000:]
    assert open(str(tmp_path / "fc_out0"), "rb").read() == open(os.path.join(golden, "fastcar_k5_u16.out"), "rb").read()
    kernels = [ln.split(": ", 1)[1] for ln in r.stderr.decode().splitlines() if ln.startswith("kernel: ")]
    assert kernels and all(kn.startswith("k_pair_tiles_multi") for kn in kernels), kernels


def _shifted(text, w0):
    """the classification block of a weights file with its intercept replaced (the line after the first `n_combos:`)"""
    lines = text.splitlines()
    at = next(i for i, ln in enumerate(lines) if ln.startswith("n_combos:")) + 1
    lines[at] = repr(float(w0))
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("dtype,k,wts,repeats", [(32, 9, "weights_k9_u32.txt", None), (32, 9, "weights_k9_u32.txt", "di"), (16, 9, "weights_cfg5_u16_k9.txt", None),
                                                 (8, 9, "weights_k9_u8.txt", "homo")])
def test_close_flags_alone_are_the_flags_of_the_fp64_evaluation(ctx, dtype, k, wts, repeats):
    """A Q x M call that wants nothing but the close flags (fastcar's Predictor::close loop, fastcar/FC_Runner.cpp:446-458; bench.py's step)
    decides them in f32 with an error bound and evaluates in FP64 only the pairs the bound leaves open (pair_features.hip, screen_close).
    The flags must be those of the FP64 evaluation pair for pair: against the same call with the sums requested (FP64 for every pair),
    against the CPU oracle, and -- with the model's intercept moved so that the weighted sums of this very set straddle 0, i.e. as many
    pairs as possible sit at the decision boundary -- against the FP64 path again. A `--feat slow` model has no f32 image: same flags
    through the unscreened kernel."""
    from oracle import oracle_py
    seqs, _ = synth.families(4242 + k + dtype, 700, 1000, family=20)
    if repeats:
        seqs = _repeat_bearing(seqs, 9, repeats)
    n = len(seqs)
    hs = api.HistogramSet(ctx, k, dtype, n)
    for off in range(0, n, 256):
        hs.build(seqs[off:off + 256], first_slot=off)
    text = weights_text(wts)
    feat = api.Feature.from_text(ctx, text, 0)
    qs = np.arange(0, n, 2, dtype=np.uint32)[:300]
    for order in (api.ORDER_CAND_FIRST, api.ORDER_QUERY_FIRST):
        full = api.score_multi(ctx, feat, hs, None, hs, qs, order=order, m=n, want=("sum", "close", "counts"))
        assert ctx.last_kernel_info()[0].startswith("k_pair_gemm_fp4_dma"), ctx.last_kernel_info()
        only = api.score_multi(ctx, feat, hs, None, hs, qs, order=order, m=n, want=("close", "counts"))
        assert ctx.last_kernel_info()[0].startswith("k_pair_gemm_fp4_dma"), ctx.last_kernel_info()
        assert np.array_equal(only["close"], full["close"]) and np.array_equal(only["counts"], full["counts"]), (wts, order)
        assert np.array_equal(full["close"], (full["sum"] >= 0).astype(np.uint8))          # (bias 0: close <=> s >= 0 for every s this far from 1e-16)
        assert int(only["close"].sum()) >= 300          # every query is close to itself
    # the oracle on a sample of rows
    pred = oracle_py.predictor(text)
    oh = [oracle_py.hist(s, k, dtype) for s in seqs]
    only = api.score_multi(ctx, feat, hs, None, hs, qs, m=n, want=("close",))
    for row in (0, 7, 150, 299):          # (candidate, query) per pair, no length window: Predictor::p_close, predict/Predictor.cpp:284-333
        f = np.array([oracle_py.score(pred.cls, oh[c], oh[int(qs[row])])[2] >= 0 for c in range(n)], dtype=np.uint8)
        assert np.array_equal(only["close"][row], f), row
    # the same set with its sums centred on the threshold: the intercept moved by the median sum
    full = api.score_multi(ctx, feat, hs, None, hs, qs, m=n, want=("sum",))
    w0 = float(text.splitlines()[next(i for i, ln in enumerate(text.splitlines()) if ln.startswith("n_combos:")) + 1])
    for quantile in (0.5, 0.02, 0.98):
        moved = api.Feature.from_text(ctx, _shifted(text, w0 - float(np.quantile(full["sum"], quantile))), 0)
        a = api.score_multi(ctx, moved, hs, None, hs, qs, m=n, want=("sum", "close"))
        b = api.score_multi(ctx, moved, hs, None, hs, qs, m=n, want=("close",))
        frac = float(a["close"].mean())
        assert abs(frac - (1 - quantile)) < 0.02, (quantile, frac)          # the boundary really runs through the set
        assert np.array_equal(a["close"], b["close"]), quantile
        assert np.array_equal(a["close"], (a["sum"] >= 0).astype(np.uint8))


def test_close_flags_alone_with_a_slow_model_and_with_a_bias(ctx):
    """models the f32 screen does not cover -- a divergence statistic, a bias (Predictor::set_bias) -- keep the FP64 evaluation: same flags"""
    seqs, _ = synth.families(991, 300, 1000, family=10)
    hs = api.HistogramSet(ctx, 9, 16, len(seqs))
    hs.build(seqs)
    qs = np.arange(130, dtype=np.uint32)
    slow = api.Feature.from_text(ctx, weights_text("weights_cfg5_k9.txt"), 0)
    a = api.score_multi(ctx, slow, hs, None, hs, qs, m=len(seqs), want=("sum", "close"))
    b = api.score_multi(ctx, slow, hs, None, hs, qs, m=len(seqs), want=("close",))
    assert np.array_equal(a["close"], b["close"])
    biased = api.Feature.from_text(ctx, weights_text("weights_k9_u8.txt"), 0)          # (a `fast` model: screened while its bias is 0)
    for bias in (0.2, -0.3, 0.0):
        biased.set_bias(bias)
        a = api.score_multi(ctx, biased, hs, None, hs, qs, m=len(seqs), want=("csum", "close"))
        b = api.score_multi(ctx, biased, hs, None, hs, qs, m=len(seqs), want=("close",))
        assert np.array_equal(a["close"], b["close"]) and np.array_equal(a["close"], (np.round(a["csum"]) > 0).astype(np.uint8)), bias
