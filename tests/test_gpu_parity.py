"""GPU parity proper: libmeshclust2_hip.so (through the C ABI) against the CPU oracle on the same seeded inputs and
against the committed golden fixtures (reference outputs). Integer work is compared bit-exact; FP64 scores to 1e-9
relative (the north-star bar is 1e-5)."""
import ctypes

import numpy as np
import pytest

from golden_util import EXACT, FEATS, VECTOR_SETS, cfg4_sequences, dense_bins, kat, load_vectors, weights_text
from meshclust2_amd import api, synth

pytestmark = pytest.mark.gpu
C_byref = ctypes.byref

# all 11 in-scope statistics (`--feat slow`); the names FAST_* are kept for the column bookkeeping below
FAST = FEATS
FAST_MASK = sum(1 << b for _, b in FEATS)
FAST_COLS = list(range(len(FEATS)))
RTOL = 1e-9


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def test_context_is_gfx950(ctx):
    assert "gfx950" in ctx.device_name()


def test_appendix_d_known_answers(ctx):
    k = kat()
    for dt in (8, 16, 32, 64):
        g = k["u%d" % dt]
        hs = api.HistogramSet(ctx, 2, dt, 2)
        hs.build([k["A"], k["B"]])
        assert hs.download(0).tolist() == g["hist_A"] and hs.download(1).tolist() == g["hist_B"]
        assert hs.info(0)["one_mers"] == g["one_mers_A"]
        raw = api.pair_features_raw(ctx, hs, [0], hs, 1, FAST_MASK, api.ORDER_CAND_FIRST)[0]
        for col, (name, _) in zip(raw, FAST):
            assert col == pytest.approx(g["raw"][name], rel=1e-13), (dt, name)


@pytest.mark.parametrize("vec,wts", VECTOR_SETS)
def test_golden_vectors(ctx, vec, wts):
    v = load_vectors(vec)
    k, dt, n = int(v["k"]), int(v["dtype"]), int(v["n"])
    hs = api.HistogramSet(ctx, k, dt, n + 2)
    hs.build([bytes(s) for s in v["seqs"]])
    for i in range(n):
        assert np.array_equal(hs.download(i), dense_bins(v, i)), i           # bit-exact k-mer counts
        inf = hs.info(i)
        assert (inf["mag"], inf["length"], inf["one_mers"]) == (int(v["mag"][i]), int(v["length"][i]), v["one_mers"][i].tolist())
        assert inf["stddev"] == pytest.approx(float(v["stddev"][i]), rel=1e-10)
    everyone = np.arange(n, dtype=np.uint32)
    for q in range(n):
        for order in (api.ORDER_CAND_FIRST, api.ORDER_QUERY_FIRST):
            raw = api.pair_features_raw(ctx, hs, everyone, hs, q, FAST_MASK, order)
            exp = v["raw"][:, q, :] if order == api.ORDER_CAND_FIRST else v["raw"][q, :, :]
            for c, col in enumerate(FAST_COLS):
                name = FEATS[col][0]
                if name in EXACT and name not in ("kulczynski2",):
                    assert np.array_equal(raw[:, c], exp[:, col]), (name, q, order)
                else:
                    assert np.allclose(raw[:, c], exp[:, col], rtol=RTOL, atol=1e-13), (name, q, order)
    feat = api.Feature.from_text(ctx, weights_text(wts), 0)
    reg = api.Feature.from_text(ctx, weights_text(wts), 1)
    for q in range(n):
        r = feat.compute(hs, everyone, hs, q, api.ORDER_CAND_FIRST)
        assert np.allclose(r["singles"], v["singles"][:, q, :], rtol=RTOL, atol=1e-12)
        assert np.allclose(r["sum"], v["sums"][:, q], rtol=1e-8, atol=1e-10)
        assert np.allclose(r["csum"], v["csums"][:, q], rtol=RTOL)
        close, sim = api.Predictor(ctx, feat, reg).search(hs, everyone, hs, q)
        assert np.array_equal(close, v["close"][:, q])
        assert np.allclose(sim, v["predict"][:, q], rtol=1e-8, atol=1e-10)
    for ci, cutoff in enumerate(v["cutoffs"]):
        trn = api.Trainer(ctx, feat, float(cutoff))
        for q in range(n):
            cands = np.array([c for c in range(n) if c != q], dtype=np.uint32)
            flags, bp, bs, im = trn.get_close(hs, cands, hs, q)
            assert np.array_equal(flags, v["get_close_flags_%d" % ci][q]), (cutoff, q)
            gbp, gbs, gim = v["get_close_best_%d" % ci][q]
            assert (bp, im) == (int(gbp), bool(gim)) and bs == pytest.approx(gbs, rel=RTOL)
            assert np.array_equal(trn.filter(hs, q, hs, cands), v["filter_%d" % ci][q]), (cutoff, q)
            if q + 1 < n:
                assert trn.merge(hs, everyone, q, q + 1, min(n - 1, q + 6)) == int(v["merge_%d" % ci][q])
    pos, d, mean = api.mean_nearest(ctx, hs, v["mean_members"].astype(np.uint32), want_mean=True)
    assert np.array_equal(mean, v["mean"]) and pos == int(v["mean_nearest"])
    assert np.allclose(d, v["mean_dist"], rtol=1e-12, atol=0)
    # Center semantics: clone re-sums mag, set() keeps the stale one
    hs.clone_from(n, hs, 0)
    hs.assign_from(n, hs, min(5, n - 1))
    assert hs.info(n)["mag"] == int(v["stale_mag"])
    assert np.array_equal(hs.download(n), dense_bins(v, min(5, n - 1)))
    raw = api.pair_features_raw(ctx, hs, [n], hs, min(3, n - 1), FAST_MASK, api.ORDER_CAND_FIRST)[0]
    assert np.allclose(raw, v["stale_raw"][FAST_COLS], rtol=RTOL, atol=1e-13)


@pytest.mark.parametrize("dtype,k,n,length", [(32, 9, 48, 1000), (16, 5, 200, 1000), (8, 9, 24, 1000), (8, 3, 16, 60), (16, 1, 8, 30),
                                              (32, 2, 12, 40), (64, 7, 12, 3000), (16, 8, 16, 5000), (32, 11, 4, 20000)])
def test_against_oracle_seeded(ctx, oracle, dtype, k, n, length):
    seqs, _ = synth.families(1000 + 7 * k + dtype, n, length, family=8)
    seqs = list(seqs) + [seqs[0][: max(length // 2, 25)], seqs[1] + seqs[2][: length // 3]]
    n = len(seqs)
    hs = api.HistogramSet(ctx, k, dtype, n)
    hs.build(seqs)
    oh = [oracle.hist(s, k, dtype) for s in seqs]
    for i in range(n):
        assert np.array_equal(hs.download(i), oh[i].array())
        inf = hs.info(i)
        assert (inf["mag"], inf["length"], inf["one_mers"], inf["overflow"]) == (oh[i].mag, oh[i].length, list(oh[i].one_mers), oh[i].overflow)
    rng = np.random.default_rng(k)
    cands = rng.permutation(n).astype(np.uint32)
    for q in (0, n - 1, n // 2):
        raw = api.pair_features_raw(ctx, hs, cands, hs, q, FAST_MASK, api.ORDER_CAND_FIRST)
        for i, c in enumerate(cands):
            for col, (name, bit) in zip(raw[i], FAST):
                exp = oracle.raw_feature(1 << bit, oh[c], oh[q])
                if name in EXACT and name != "kulczynski2":
                    assert col == exp, (name, c, q)
                else:
                    assert col == pytest.approx(exp, rel=RTOL, abs=1e-13), (name, c, q)
    # identity m = None (slots 0..m-1) path
    raw2 = api.pair_features_raw(ctx, hs, None, hs, 0, FAST_MASK, api.ORDER_CAND_FIRST, m=n)
    raw3 = api.pair_features_raw(ctx, hs, np.arange(n, dtype=np.uint32), hs, 0, FAST_MASK, api.ORDER_CAND_FIRST)
    assert np.array_equal(raw2, raw3)
    mem = np.arange(0, n, 2, dtype=np.uint32)
    pos, d, mean = api.mean_nearest(ctx, hs, mem, want_mean=True)
    om, od, onear = oracle.mean_nearest([oh[i] for i in mem])
    assert np.array_equal(mean, om) and pos == onear and np.allclose(d, od, rtol=1e-12, atol=0)


def test_saturation_and_overflow_flag(ctx, oracle):
    seqs = [b"A" * 700 + b"C" * 30, b"ACGT" * 200, b"AC" * 40000]
    for dtype, k in ((8, 3), (16, 2), (8, 6)):
        hs = api.HistogramSet(ctx, k, dtype, len(seqs))
        hs.build(seqs)
        for i, s in enumerate(seqs):
            o = oracle.hist(s, k, dtype)
            assert np.array_equal(hs.download(i), o.array())
            assert hs.info(i)["overflow"] == o.overflow
            assert hs.info(i)["mag"] == o.mag


def test_encoding_edge_cases_through_build(ctx, oracle):
    nasty = [
        b"ACGTNNNNACGTACGTACGTACGTAACCGGTTNNNNNNNNNNNNACGATCGATCGATCGATCGACTAGCTAGCTAGCATCGAT",
        b"acgtacgtnnacgtRYMKSWHBVDacgtacgtacgtagctagcatcgatcgatcgatcagctagcat",
        b"NNNNNNNNNNNNNNNNNNNNNNNNNNNNNN", b"", b"ACG",
        b"ACGTACGTACGTACGTACGTACGTNNNNNNNNNNNNNNNNNNNNNNA",
        b"ACGTACGTACGTACGTACGTNNNNNNNNNNNNNNNACGTACGTACGTAC",
        b"ACGTACGTACGTACGTACGTACGNNNNNNNNNACGTACGTACGTACGTACGTACGT",
        b"ACGTNNACGTNNACGTNN",
    ]
    for strip in (False, True):
        hs = api.HistogramSet(ctx, 4, 16, len(nasty))
        hs.build(nasty, strip=strip)
        for i, s in enumerate(nasty):
            o = oracle.hist(s, 4, 16, strip)
            assert np.array_equal(hs.download(i), o.array()), (i, strip)
            assert hs.info(i)["length"] == o.length
    for s in nasty:
        assert api.encode(s) == oracle.encode(s)
    with pytest.raises(api.MscError) as e:
        api.HistogramSet(ctx, 4, 16, 1).build([b"ACGTACGTACGTACGTACGTAC-GTACGTACGTACGTACGT"])
    assert e.value.code == -5


def test_zero_length_point_is_an_error(ctx):
    txt = weights_text("weights_k5_u16.txt")
    feat = api.Feature.from_text(ctx, txt, 0)
    hs = api.HistogramSet(ctx, 5, 16, 2)
    hs.build([b"ACGT" * 50, b"N" * 40])
    with pytest.raises(api.MscError) as e:
        feat.compute(hs, [1], hs, 0)
    assert e.value.code == -6        # the reference throws 123 (predict/Feature.cpp:878-886)


def test_unsupported_inputs_fail_loudly(ctx):
    with pytest.raises(api.MscError):
        api.HistogramSet(ctx, 14, 32, 1)
    hs = api.HistogramSet(ctx, 3, 32, 2)
    hs.build([b"ACGT" * 10, b"ACGT" * 10])
    with pytest.raises(api.MscError):
        api.pair_features_raw(ctx, hs, [0], hs, 1, 1 << 1)          # hellinger: out of scope (extraslow only)
    with pytest.raises(api.MscError):
        api.Feature.create(ctx, 3, [(0, 1 << 12)], [0.0, 1.0], [(1 << 12, 0.0, 1.0)])    # markov


G1_WEIGHTS = """k: 9
mode: 1
max_features: 4
ID: 0.9
Datatype: uint32_t
feature_set: 81920

n_combos: 3
-0.4
0 65536 1.7
1 81920 -0.9
3 16384 0.35

n_singles: 2
65536 0 0.02
16384 0 0.5
"""


@pytest.mark.parametrize("dtype,k,length,layout", [(32, 9, 1000, "dense"), (32, 9, 1000, "sparse"), (8, 9, 700, "dense"), (16, 8, 3000, "dense"),
                                                     (64, 10, 2500, "sparse"), (16, 11, 9000, "sparse"),
                                                     # histograms under 64 KiB: no list form, the dense group kernels
                                                     (16, 5, 300, "dense"), (8, 7, 900, "dense"), (32, 6, 600, "dense"), (64, 3, 120, "dense"),
                                                     (64, 4, 200, "dense"), (8, 2, 90, "dense")])
def test_sim_mm_and_rre_k_r_against_the_oracle(ctx, oracle, dtype, k, length, layout):
    """The two `extraslow` statistics BASELINE's north_star names: sim_mm (through Feature<T>::markov / d_markov) and rre_k_r
    (predict/Feature.cpp:1367-1393,1429-1455,1029-1062), sums over the groups of four bins that share a (k-1)-mer prefix. Scored by
    the merge kernels' group pass (a dense set through its sparse mirror; histograms under 64 KiB from the dense slots): raw values, both argument orders, a model built on
    them, get_close / filter decisions -- against the oracle, which tests/test_oracle_vs_ref.py pins to the reference's own
    static functions. The stale magnitude of a moved centre enters through getRealMagnitude."""
    seqs, _ = synth.families(9100 + k + dtype, 18, length, family=6, length_jitter=length // 10)
    seqs = list(seqs) + [seqs[0], b"ACGGT" * 60 + seqs[1][:400]]
    n = len(seqs)
    hs = api.HistogramSet(ctx, k, dtype, n + 1, sparse_entries=(sum(len(s_) for s_ in seqs) * 2 + 4096) if layout == "sparse" else 0)
    hs.build(seqs)
    oh = [oracle.hist(s_, k, dtype) for s_ in seqs]
    mask = (1 << 14) | (1 << 16) | (1 << 2) | (1 << 29)          # with a fast statistic and a divergence in the same call
    cands = np.arange(n, dtype=np.uint32)[::-1].copy()
    for q in (0, 7, n - 1):
        for order in (api.ORDER_CAND_FIRST, api.ORDER_QUERY_FIRST):
            raw = api.pair_features_raw(ctx, hs, cands, hs, q, mask, order)
            for i, c in enumerate(cands):
                a, b = (oh[c], oh[q]) if order == api.ORDER_CAND_FIRST else (oh[q], oh[c])
                assert raw[i][0] == oracle.raw_feature(1 << 2, a, b)
                for col, bit in ((1, 14), (2, 16), (3, 29)):
                    exp = oracle.raw_feature(1 << bit, a, b)
                    assert raw[i][col] == pytest.approx(exp, rel=1e-9, abs=1e-13), (q, c, bit, order)
    assert api.pair_features_raw(ctx, hs, [n - 2], hs, 0, 1 << 16)[0][0] == 0.0          # identical sequences: sim_mm = 1 - exp(0)
    feat = api.Feature.from_text(ctx, G1_WEIGHTS, 0)
    pred = oracle.predictor(G1_WEIGHTS)
    r = feat.compute(hs, cands, hs, 3)
    for i, c in enumerate(cands):
        s_, _, wsum = oracle.score(pred.cls, oh[c], oh[3])
        assert np.allclose(r["singles"][i], s_, rtol=1e-8, atol=1e-11) and r["sum"][i] == pytest.approx(wsum, rel=1e-8, abs=1e-10)
    trn = api.Trainer(ctx, feat, 0.9)
    w = np.array([c for c in range(n) if c != 2], dtype=np.uint32)
    flags, bp, bs, im = trn.get_close(hs, w, hs, 2)
    of, obp, obs, oim = oracle.get_close(pred, 0.9, oh[2], [oh[c] for c in w])
    assert np.array_equal(flags, of) and (bp, im) == (obp, oim) and bs == pytest.approx(obs, rel=1e-8)
    assert np.array_equal(trn.filter(hs, 2, hs, w), oracle.filter_(pred, 0.9, oh[2], [oh[c] for c in w]))
    # Q x M: the group passes queued behind the streaming kernel (dense sets) / one 1 x M pass per query (sparse sets) -- same records
    qs = [3, 5, 0, n - 1, 7, 2, 9, 11]
    multi = api.score_multi(ctx, feat, hs, cands, hs, qs, feat_mask=(1 << 14) | (1 << 16))
    assert np.array_equal(multi["sum"][0], r["sum"])
    for qi, q in enumerate(qs):
        one = api.pair_features_raw(ctx, hs, cands, hs, q, (1 << 14) | (1 << 16))
        assert np.array_equal(np.asarray(multi["raw"][qi]).reshape(len(cands), 2), np.asarray(one).reshape(len(cands), 2)), q
        assert np.array_equal(multi["sum"][qi], feat.compute(hs, cands, hs, q)["sum"]), q
    # a moved centre keeps its stale magnitude: clone of 4, then set(9)
    hs.clone_from(n, hs, 4)
    hs.assign_from(n, hs, 9)
    oc = oracle.Hist()
    oracle.lib().orc_hist_clone(C_byref(oh[4]), C_byref(oc))
    oracle.lib().orc_hist_set(C_byref(oc), C_byref(oh[9]))
    got = api.pair_features_raw(ctx, hs, [n], hs, 5, (1 << 14) | (1 << 16))[0]
    assert got[0] == pytest.approx(oracle.raw_feature(1 << 14, oc, oh[5]), rel=1e-9) and got[1] == pytest.approx(oracle.raw_feature(1 << 16, oc, oh[5]), rel=1e-9)
    for h in oh:
        oracle.lib().orc_hist_free(h)


def test_group_statistics_dense_and_list_forms_agree(ctx):
    """the dense group kernels and the list form use the same per-group terms over the same 16 index sub-ranges: the same histograms
    held as a 64 KiB dense set (list form through the mirror) and scored with the mirror refused (dense form, in a child process:
    the switch is read once) give the same bits"""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import numpy as np, sys\n"
        "from meshclust2_amd import api, synth\n"
        "ctx = api.Context(0)\n"
        "seqs, _ = synth.families(515, 12, 2000, family=4, length_jitter=100)\n"
        "hs = api.HistogramSet(ctx, 8, 16, len(seqs)); hs.build(list(seqs))\n"
        "r = api.pair_features_raw(ctx, hs, np.arange(len(seqs), dtype=np.uint32), hs, 3, (1 << 14) | (1 << 16))\n"
        "sys.stdout.write(np.asarray(r, dtype=np.float64).tobytes().hex())\n")
    outs = []
    for env_extra in ({}, {"MSC_NO_SPARSE_MIRROR": "1"}):
        env = dict(os.environ, **env_extra)
        env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
        assert p.returncode == 0, p.stderr
        outs.append(p.stdout.strip())
    assert outs[0] == outs[1] and len(outs[0]) == 12 * 2 * 16


@pytest.mark.parametrize("layout", ["dense", "sparse"])
def test_copy_batch_is_an_exact_copy(ctx, layout):
    """msc_hist_copy_batch (relocating / compacting a centre store in one launch): bins, list and every word of the record -- a moved
    centre's stale magnitude included -- equal the slot-by-slot msc_hist_copy, and scores against the copies are the same bits"""
    seqs, _ = synth.families(4242, 12, 1500, family=4, length_jitter=200)
    n = len(seqs)
    k, dtype = 8, 16
    ent = (sum(len(s_) for s_ in seqs) * 3 + 4096) if layout == "sparse" else 0
    pts = api.HistogramSet(ctx, k, dtype, n, sparse_entries=ent)
    pts.build(list(seqs))
    assert np.array_equal(pts.lengths(), [pts.info(i)["length"] for i in range(n)]) and np.array_equal(pts.lengths(3, 4), pts.lengths()[3:7])      # msc_hist_lengths
    centres = api.HistogramSet(ctx, k, dtype, n, sparse_entries=ent)
    for i in range(n):
        centres.clone_from(i, pts, i)
    for i in range(0, n, 2):                       # moved centres: set() keeps the magnitude of the histogram they were cloned from
        centres.assign_from(i, pts, (i + 5) % n)
    one = api.HistogramSet(ctx, k, dtype, n + 3, sparse_entries=ent)
    for i in range(n):
        one.copy_from(i, centres, i)
    order = np.array([7, 0, 3, 11, 1, 2, 9, 4, 10, 5, 8, 6], dtype=np.uint32)      # any distinct destinations
    many = api.HistogramSet(ctx, k, dtype, n + 3, sparse_entries=ent)
    many.copy_batch(order, centres, np.arange(n, dtype=np.uint32))
    for i in range(n):
        a, b, c = centres.info(i), one.info(i), many.info(int(order[i]))
        assert a == b == c
        assert np.array_equal(one.download(i), many.download(int(order[i]))) and np.array_equal(one.download(i), centres.download(i))
    assert any(centres.info(i)["mag"] != centres.info(i)["sum"] for i in range(0, n, 2))          # the stale magnitudes are really there
    mask = FAST_MASK
    r1 = api.pair_features_raw(ctx, one, np.arange(n, dtype=np.uint32), pts, 3, mask)
    r2 = api.pair_features_raw(ctx, many, order, pts, 3, mask)
    assert np.array_equal(np.asarray(r1), np.asarray(r2))
    # msc_hist_clone_batch == msc_hist_clone slot by slot (magnitude re-summed from the bins)
    c1 = api.HistogramSet(ctx, k, dtype, n, sparse_entries=ent)
    c2 = api.HistogramSet(ctx, k, dtype, n, sparse_entries=ent)
    for i in range(n):
        c1.clone_from(int(order[i]), centres, i)
    c2.clone_batch(order, centres, np.arange(n, dtype=np.uint32))
    for i in range(n):
        assert c1.info(i) == c2.info(i) and c1.info(i)["mag"] == c1.info(i)["sum"] and np.array_equal(c1.download(i), c2.download(i))
    if layout == "sparse":                          # all or nothing against the entry arena
        tiny = api.HistogramSet(ctx, k, dtype, n, sparse_entries=64)
        with pytest.raises(api.MscError, match="arena"):
            tiny.copy_batch(np.arange(n, dtype=np.uint32), centres, np.arange(n, dtype=np.uint32))
        # msc_hist_set_clear: the arena of a filled store is free again, and what is copied in afterwards is what a fresh store holds
        exact = sum(centres.entries(i) for i in range(n))
        snug = api.HistogramSet(ctx, k, dtype, n + 3, sparse_entries=exact + exact // 2)
        snug.copy_batch(order, centres, np.arange(n, dtype=np.uint32))
        with pytest.raises(api.MscError, match="arena"):          # (append-only: a second round does not fit ...
            snug.copy_batch(order, centres, np.arange(n, dtype=np.uint32))
        snug.clear()                                              # ... until the store is emptied)
        assert all(snug.entries(i) == 0 for i in range(n + 3))
        snug.copy_batch(order, centres, np.arange(n, dtype=np.uint32))
        for i in range(n):
            assert snug.info(int(order[i])) == many.info(int(order[i])) and np.array_equal(snug.download(int(order[i])), many.download(int(order[i])))
        r3 = api.pair_features_raw(ctx, snug, order, pts, 3, mask)
        assert np.array_equal(np.asarray(r1), np.asarray(r3))
    else:
        with pytest.raises(api.MscError, match="sparse sets only"):
            many.clear()


def test_batch_slot_entry_points_reject_bad_arguments(ctx):
    """msc_hist_copy_batch / _clone_batch / _lengths: out-of-range slots, mismatched sets and empty batches"""
    a = api.HistogramSet(ctx, 5, 16, 4)
    b = api.HistogramSet(ctx, 5, 16, 4)
    c = api.HistogramSet(ctx, 6, 16, 4)
    a.build([b"ACGTTGCAAC" * 20, b"TTGACCA" * 30, b"GGGATCCA" * 25, b"ACGT" * 40])
    b.copy_batch(np.zeros(0, dtype=np.uint32), a, np.zeros(0, dtype=np.uint32))          # nothing to do
    with pytest.raises(api.MscError, match="slot out of range"):
        b.copy_batch([0, 4], a, [0, 1])
    with pytest.raises(api.MscError, match="slot out of range"):
        b.clone_batch([0, 1], a, [0, 9])
    with pytest.raises(api.MscError, match="differ"):
        c.copy_batch([0], a, [0])
    b.clone_batch([3, 1], a, [0, 2])
    assert np.array_equal(b.download(3), a.download(0)) and np.array_equal(b.download(1), a.download(2))
    assert list(a.lengths()) == [200, 210, 200, 160] and list(b.lengths(1, 1)) == [200]
    with pytest.raises(api.MscError):
        a.lengths(2, 3)                                                                    # runs past the capacity


def test_upload_round_trip_and_properties(ctx):
    """Size-independent properties: symmetric statistics, self-pair identities, checksum of checksums."""
    rng = np.random.default_rng(3)
    k, dt, n = 7, 32, 6
    hs = api.HistogramSet(ctx, k, dt, n)
    data = [rng.integers(1, 40, size=4 ** k).astype(np.uint32) for _ in range(n)]
    for i, b in enumerate(data):
        hs.upload(i, b, 1000 + i)
        assert np.array_equal(hs.download(i), b)
        assert hs.info(i)["sum"] == int(b.sum()) and hs.info(i)["sum_sq"] == int((b.astype(np.uint64) ** 2).sum())
    ev = np.arange(n, dtype=np.uint32)
    m = np.stack([api.pair_features_raw(ctx, hs, ev, hs, q, FAST_MASK & ~(1 << 28)) for q in range(n)])    # [q, c, f]
    names = [nm for nm, b in FAST if b != 28]
    div = [names.index("jefferey_divergence"), names.index("jensen_shannon")]
    rest = [i for i in range(len(names)) if i not in div]
    assert np.array_equal(m[:, :, rest], m.transpose(1, 0, 2)[:, :, rest])      # symmetric (simratio/u32 excluded: SURVEY Q3)
    assert np.allclose(m[:, :, div], m.transpose(1, 0, 2)[:, :, div], rtol=1e-12, atol=0)
    for q in range(n):
        row = dict(zip(names, m[q, q]))
        assert row["manhattan"] == 0 and row["euclidean"] == 0 and row["emd"] == 0 and row["intersection"] == 1.0
        assert row["normalized_vectors"] == pytest.approx(1.0, abs=1e-15) and row["pearson"] == pytest.approx(1.0, abs=1e-12)
        assert row["jefferey_divergence"] == 0 and row["jensen_shannon"] == 0
    for q in range(n):
        for c in range(n):
            d = data[q].astype(np.int64) - data[c].astype(np.int64)
            assert m[q, c, names.index("manhattan")] == np.abs(d).sum()
            assert m[q, c, names.index("emd")] == np.abs(np.cumsum(d)).sum()


_TORCH_VIEW_SCRIPT = r"""
import sys
import numpy as np
import torch                      # torch first: one HIP runtime per process, like bench.py
sys.path.insert(0, sys.argv[1])
from meshclust2_amd import api, synth
ctx = api.Context(0)
seqs, _ = synth.families(3, 4, 500, family=4)
hs = api.HistogramSet(ctx, 6, 16, 5)
hs.build(seqs)
bins_ptr, slot_bytes, scal_ptr, scal_bytes = hs.device_view()
class View:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}
b = torch.as_tensor(View(bins_ptr, slot_bytes * 5), device="cuda")
s = torch.as_tensor(View(scal_ptr, scal_bytes * 5), device="cuda")
b[4 * slot_bytes:5 * slot_bytes].copy_(b[2 * slot_bytes:3 * slot_bytes])
s[4 * scal_bytes:5 * scal_bytes].copy_(s[2 * scal_bytes:3 * scal_bytes])
torch.cuda.synchronize()
hs.import_done(4, 1)
assert np.array_equal(hs.download(4), hs.download(2)) and hs.info(4) == hs.info(2)
mask = sum(1 << x for x in (2, 3, 5, 9, 13, 18, 21, 27, 28))
a = api.pair_features_raw(ctx, hs, [0, 1, 3], hs, 4, mask)
c = api.pair_features_raw(ctx, hs, [0, 1, 3], hs, 2, mask)
assert np.array_equal(a, c)
print("TORCH_VIEW_OK")
"""


def test_device_view_is_usable_from_torch():
    """The multi-GPU path hands raw device regions to torch.distributed (RCCL) through __cuda_array_interface__:
    copy a slot with torch, call import_done, and the library must see the new histogram."""
    import os
    import subprocess
    import sys
    pytest.importorskip("torch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _TORCH_VIEW_SCRIPT, root], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert b"TORCH_VIEW_OK" in out.stdout, out.stdout.decode(errors="replace")[-2000:]


@pytest.mark.parametrize("dtype,k,nq", [(32, 9, 5), (16, 5, 7), (8, 9, 4), (16, 7, 2), (64, 6, 3), (8, 3, 3), (16, 9, 6), (8, 6, 9), (16, 6, 19), (8, 8, 16), (32, 5, 8), (32, 6, 17), (32, 7, 20)])
def test_multi_query_pass_equals_single_query_passes(ctx, dtype, k, nq):
    """msc_score_multi (candidate tiles reused across several query tiles) == nq independent 1 x M passes, bit for bit."""
    seqs, _ = synth.families(500 + k, 40, 1000 if k > 3 else 60, family=10)
    hs = api.HistogramSet(ctx, k, dtype, len(seqs))
    hs.build(seqs)
    wts = "weights_k9_u32.txt" if dtype == 32 else "weights_k5_u16_slow.txt" if k == 7 else "weights_k5_u16.txt"
    feat = api.Feature.from_text(ctx, weights_text(wts), 0)
    cands = np.arange(3, len(seqs), dtype=np.uint32)
    qs = np.arange(nq, dtype=np.uint32) * 2
    mask = FAST_MASK if k == 7 else (FAST_MASK & ~((1 << 7) | (1 << 29)))
    multi = api.score_multi(ctx, feat, hs, cands, hs, qs, feat_mask=mask)
    for i, q in enumerate(qs):
        single = feat.compute(hs, cands, hs, int(q))
        raw = api.pair_features_raw(ctx, hs, cands, hs, int(q), mask)
        assert np.array_equal(multi["sum"][i], single["sum"]) and np.array_equal(multi["csum"][i], single["csum"])
        assert np.array_equal(multi["raw"][i], raw)
        assert np.array_equal(multi["close"][i], (np.round(single["csum"]) > 0).astype(np.uint8))


@pytest.mark.parametrize("dtype,k,nq,length", [(32, 9, 16, 1000), (16, 9, 16, 1000), (8, 9, 20, 1000), (16, 8, 5, 3000), (32, 8, 3, 2000), (64, 9, 4, 800)])
def test_divergence_statistics_are_the_same_in_every_route(ctx, oracle, dtype, k, nq, length):
    """jefferey_divergence / jensen_shannon are FP64 sums over 4^k bins; every route scores them with the ONE merge kernel over
    sorted (bin, value) lists (a dense set through its sparse mirror), so the dense 1 x M pass, the dense Q x M
    pass (which no longer falls back to one k_pair_tiles pass per query), the batched update stage and a sparse set of the same
    sequences return BIT-IDENTICAL values and decisions -- tied candidates stay tied whatever the route. Against the oracle the
    values hold 1e-9 (predict/Feature.cpp:984-1009,1231-1263). Includes a tandem repeat whose counts pass the 16 x 16 term table
    and (8-bit) saturate, and rewritten slots (the mirror refreshes what was written)."""
    seqs, _ = synth.families(7300 + k + dtype, 44, length, family=11, length_jitter=length // 8)
    seqs = list(seqs)
    seqs[5] = seqs[5][:200] + b"ACG" * 130 + seqs[5][200:]          # counts ~130 for three k-mers (past the 16 x 16 term table)
    seqs[6] = seqs[6][:100] + b"AC" * 300 + seqs[6][100:]           # counts ~300: saturates uint8_t bins
    n = len(seqs)
    dense = api.HistogramSet(ctx, k, dtype, n + 2)
    sparse = api.HistogramSet(ctx, k, dtype, n + 2, sparse_entries=sum(len(s_) for s_ in seqs) * 2 + 4096)
    dense.build(seqs)
    sparse.build(seqs)
    slow = api.Feature.from_text(ctx, weights_text("weights_cfg5_k9.txt"), 0)          # the reference-trained cfg5 model: uses jensen_shannon
    div_mask = (1 << 7) | (1 << 29)
    mask = FAST_MASK
    cands = np.arange(n, dtype=np.uint32)[::-1].copy()
    qs = (np.arange(nq, dtype=np.uint32) * 3) % n
    multi = api.score_multi(ctx, slow, dense, cands, dense, qs, feat_mask=mask)
    if nq >= 8:
        assert "k_pair_tiles" not in ctx.last_kernel_info()[0], ctx.last_kernel_info()          # the digest kernel, not one 1 x M pass per query
    oh = {int(q): oracle.hist(seqs[int(q)], k, dtype) for q in qs[:3]}
    ohc = [oracle.hist(seqs[int(c)], k, dtype) for c in cands]
    for i, q in enumerate(qs):
        one = slow.compute(dense, cands, dense, int(q))
        raw_d = api.pair_features_raw(ctx, dense, cands, dense, int(q), mask)
        raw_s = api.pair_features_raw(ctx, sparse, cands, sparse, int(q), mask)
        sp = slow.compute(sparse, cands, sparse, int(q))
        assert np.array_equal(multi["raw"][i], raw_d), (i, "Q x M vs 1 x M")
        assert np.array_equal(raw_d, raw_s), (i, "dense vs sparse")
        assert np.array_equal(multi["sum"][i], one["sum"]) and np.array_equal(one["sum"], sp["sum"]), i
        assert np.array_equal(multi["close"][i], (np.round(one["csum"]) > 0).astype(np.uint8))
        if int(q) in oh:
            for j, c in enumerate(cands):
                for bit, col in ((7, 3), (29, 10)):
                    assert raw_d[j][col] == pytest.approx(oracle.raw_feature(1 << bit, ohc[j], oh[int(q)]), rel=1e-9, abs=1e-14), (q, c, bit)
    # both argument orders
    a = api.pair_features_raw(ctx, dense, cands, dense, 2, div_mask, api.ORDER_QUERY_FIRST)
    b = api.pair_features_raw(ctx, sparse, cands, sparse, 2, div_mask, api.ORDER_QUERY_FIRST)
    assert np.array_equal(a, b)
    # the operators: window, filter, batched update == per-centre calls == the sparse layout
    for cutoff in (0.9, 0.6):
        td, ts = api.Trainer(ctx, slow, cutoff), api.Trainer(ctx, slow, cutoff)
        w = np.array([c for c in range(n) if c != 4], dtype=np.uint32)
        f1, bp1, bs1, im1 = td.get_close(dense, w, dense, 4)
        f2, bp2, bs2, im2 = ts.get_close(sparse, w, sparse, 4)
        assert np.array_equal(f1, f2) and (bp1, bs1, im1) == (bp2, bs2, im2)
    # rewritten slots: the mirror follows (clone + assign into the tail slots, then an in-place rebuild of slot 0)
    for hs in (dense, sparse):
        hs.clone_from(n, hs, 5)
        hs.assign_from(n, hs, 6)
        hs.clone_from(n + 1, hs, 7)
        hs.build([seqs[9]], first_slot=0)
    tail = np.array([n, n + 1, 0, 9], dtype=np.uint32)
    a = api.pair_features_raw(ctx, dense, tail, dense, n, div_mask)
    b = api.pair_features_raw(ctx, sparse, tail, sparse, n, div_mask)
    assert np.array_equal(a, b) and a[0][0] == 0.0 and np.array_equal(a[2], a[3])
    multi = api.score_multi(ctx, slow, dense, tail, dense, np.array([n, 0, n + 1], dtype=np.uint32), feat_mask=div_mask)
    assert np.array_equal(multi["raw"][0], a)
    for h in list(oh.values()) + ohc:
        oracle.lib().orc_hist_free(h)


@pytest.mark.parametrize("tq,slots,p16", [(4, 2, 1), (4, 3, 0), (4, 3, 1), (4, 4, 1), (8, 2, 0), (8, 3, 1), (8, 4, 0), (8, 4, 1)])
def test_multi_ring_variants(tq, slots, p16):
    """The LDS-DMA ring form of the Q x M kernel (query tile 4 or 8, 2-4 ring slots per wave, the deepest one past 64 KiB of
    LDS per workgroup) == independent 1 x M passes, bit for bit, including padded query groups, reversed slot lists and
    windows shorter than the ring is deep; with the packed 16-bit prefix form (p16) and the 32-bit one."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MSC_MULTI_TQ=str(tq), MSC_RING_SLOTS=str(slots))
    env.pop("MSC_MULTI_NO_RING", None)
    env.pop("MSC_RING_NO_P16", None)
    if not p16:
        env["MSC_RING_NO_P16"] = "1"
    out = subprocess.run([sys.executable, os.path.join(here, "ring_variant_check.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert b"RING_VARIANT_OK" in out.stdout, out.stdout.decode(errors="replace")[-3000:]


@pytest.mark.parametrize("slots", [2, 3, 4, 6, 8])
def test_multi_digest_variants(slots):
    """The digest form of the Q x M kernel (pair_digest.hip: 16 queries per workgroup, one LDS copy of each candidate tile
    shared by four waves, ring of 2-8 slots) == independent 1 x M passes, bit for bit: 8- and 16-bit count forms, padded
    query groups, slot lists, windows shorter than the ring, stale-digest refresh after slots are overwritten."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MSC_DIGEST_SLOTS=str(slots))
    for k in ("MSC_MULTI_TQ", "MSC_MULTI_NO_DIGEST", "MSC_MULTI_NO_RING", "MSC_RING_NO_P16", "MSC_RING_SLOTS"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(here, "ring_variant_check.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert b"RING_VARIANT_OK" in out.stdout, out.stdout.decode(errors="replace")[-3000:]


@pytest.mark.parametrize("switches,kernel", [("", "k_pair_gemm_fp4_dma<64 query rows"), ("MSC_MULTI_NO_GEMM", "k_pair_digest_multi"),
                                             ("MSC_MULTI_NO_GEMM MSC_MULTI_NO_RANKS", "k_pair_digest_multi"), ("MSC_MULTI_NO_RANKS", "k_pair_digest_multi"),
                                             ("MSC_GEMM_SLICES=8", "k_pair_gemm_fp4_dma<64 query rows"), ("MSC_NO_RANKS16", "k_pair_gemm_fp4_dma<64 query rows"),
                                             ("MSC_GEMM_NO_PIPE MSC_NO_SCREEN", "k_pair_gemm_fp4_dma<64 query rows"), ("MSC_GEMM_NO_PREP MSC_GEMM_NO_QUEUE", "k_pair_gemm_fp4_dma<64 query rows")])
def test_multi_route_variants(switches, kernel):
    """Every route of the Q x M pass over a dense set == independent 1 x M passes, bit for bit (the library reads its switches once
    per process), the kernel each set of switches must select asserted by name: everything on the matrix cores (one FP4 product per tile
    of presence bits + corrections from the lists of large bins, msc_pair_gemm.hip) with the earth mover's distance from 16-bit reduced
    ranks (msc_emd_ranks.hip); the digest kernel with its own products + ranks; the r02 digest kernel alone; without the ranks mirror (the
    matrix-core pass then only serves models without emd); the matrix-core pass cut into eight slices of the bins; the 32-bit rank walk;
    one stream and FP64 flags; the queries' side on the product's stream and a host wait per block."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ)
    for k in ("MSC_MULTI_TQ", "MSC_MULTI_NO_DIGEST", "MSC_MULTI_NO_RING", "MSC_RING_NO_P16", "MSC_RING_SLOTS", "MSC_DIGEST_SLOTS",
              "MSC_MULTI_NO_RANKS", "MSC_MULTI_NO_GEMM", "MSC_DIGEST_NO_TQ8", "MSC_GEMM_SLICES", "MSC_NO_RANKS16", "MSC_GEMM_NO_PIPE", "MSC_NO_SCREEN", "MSC_GEMM_NO_PREP", "MSC_GEMM_NO_QUEUE"):
        env.pop(k, None)
    for sw in switches.split():
        name, _, val = sw.partition("=")
        env[name] = val or "1"
    env["MSC_TEST_EXPECT_KERNEL"] = kernel
    out = subprocess.run([sys.executable, os.path.join(here, "ring_variant_check.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert b"RING_VARIANT_OK" in out.stdout, out.stdout.decode(errors="replace")[-3000:]


def test_cluster_driver_reproduces_reference_clstr(tmp_path):
    """the from-scratch mean-shift driver over the GPU path, fed the model the reference trained, writes
    the SAME .clstr bytes as the reference CLI did for cfg1 (1000 x 1 kb, --id 0.9 --kmer 5 --datatype 16, 1 thread)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "meshclust2_amd", "host", "msc_cluster")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "meshclust2_amd", "host")])
    seqs, hdrs = synth.families(20260001, 1000, 1000)
    fa = str(tmp_path / "cfg1.fa")
    synth.write_fasta(fa, seqs, hdrs)
    out = str(tmp_path / "out.clstr")
    golden = os.path.join(root, "tests", "golden")
    r = subprocess.run([exe, fa, "--recover", os.path.join(golden, "weights_k5_u16.txt"), "--id", "0.9", "--kmer", "5", "--datatype", "16", "--output", out],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
    got, exp = open(out, "rb").read(), open(os.path.join(golden, "cfg1.clstr"), "rb").read()
    assert got == exp, "CLSTR differs: %d vs %d bytes\n%s" % (len(got), len(exp), r.stdout.decode(errors="replace")[-500:])


def test_cluster_driver_mixed_lengths(tmp_path):
    """Same, on 2400 sequences in three length groups (--id 0.8 --kmer 6): several bvec bins, length windows that prune,
    the inclusive-end window quirk and stale-magnitude centres (Q7) all in play."""
    import os
    import subprocess
    sys_path_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(sys_path_root, "meshclust2_amd", "host", "msc_cluster")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(sys_path_root, "meshclust2_amd", "host")])
    seqs, hdrs = [], []
    for gi, (n, length, seed) in enumerate(((800, 600, 31), (800, 1000, 32), (800, 1500, 33))):
        s_, h_ = synth.families(seed, n, length, length_jitter=120)
        seqs += s_
        hdrs += [">m%d_%s" % (gi, x[1:]) for x in h_]
    fa = str(tmp_path / "mixed.fa")
    synth.write_fasta(fa, seqs, hdrs)
    out = str(tmp_path / "out.clstr")
    golden = os.path.join(sys_path_root, "tests", "golden")
    r = subprocess.run([exe, fa, "--recover", os.path.join(golden, "weights_mixed_k6_u16.txt"), "--id", "0.8", "--kmer", "6", "--datatype", "16", "--output", out],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
    got, exp = open(out, "rb").read(), open(os.path.join(golden, "mixed.clstr"), "rb").read()
    assert got == exp, "CLSTR differs: %d vs %d bytes" % (len(got), len(exp))


@pytest.mark.parametrize("dtype,k,hi", [(64, 5, 3_000_000), (32, 6, 60_000), (16, 5, 60_000), (64, 7, 9_000)])
def test_wide_counts_take_the_64bit_kernel(ctx, oracle, dtype, k, hi):
    """Bins above the 32-bit fast path's range (> 8191) run through k_pair_tiles_wide. With 64-bit bins the reference's
    arithmetic is exact, so every statistic is compared with the oracle; for narrower bins the statistics whose
    reference accumulators stay exact (manhattan, intersection, emd, kulczynski2, pearson, divergences) are."""
    import ctypes as C
    rng = np.random.default_rng(k + dtype)
    n = 6
    hs = api.HistogramSet(ctx, k, dtype, n + 1)
    data = []
    for i in range(n):
        b = rng.integers(1, 40, size=4 ** k).astype(np.uint64)
        b[rng.integers(0, 4 ** k, size=25)] = rng.integers(hi // 2, hi, size=25)
        data.append(b.astype(api.NP_T[dtype]))
        hs.upload(i, data[-1], 5000 + 37 * i)
        assert np.array_equal(hs.download(i), data[-1])

    def ohist(bins, length):
        h = oracle.Hist()
        h.dtype, h.k, h.nbins = dtype, k, 4 ** k
        h._keep = np.ascontiguousarray(bins)
        h.bins = h._keep.ctypes.data_as(C.c_void_p).value
        h.mag, h.length = int(bins.astype(np.uint64).sum()), length
        return h
    oh = [ohist(d, 5000 + 37 * i) for i, d in enumerate(data)]
    exact_ok = {"manhattan", "intersection", "emd", "length_difference", "kulczynski2"}
    approx_ok = {"pearson", "jefferey_divergence", "jensen_shannon"}
    cands = np.arange(n, dtype=np.uint32)
    for q in range(n):
        raw = api.pair_features_raw(ctx, hs, cands, hs, q, FAST_MASK)
        for c in range(n):
            for col, (name, bit) in zip(raw[c], FEATS):
                exp = oracle.raw_feature(1 << bit, oh[c], oh[q])
                if dtype == 64 or name in exact_ok:
                    if name in EXACT and name != "kulczynski2":
                        assert col == exp, (name, c, q)
                    else:
                        assert col == pytest.approx(exp, rel=1e-9, abs=1e-13), (name, c, q)
                elif name in approx_ok:
                    assert col == pytest.approx(exp, rel=1e-9, abs=1e-13), (name, c, q)
    pos, d, _ = api.mean_nearest(ctx, hs, cands)
    _, od, onear = oracle.mean_nearest(oh)
    assert pos == onear and np.allclose(d, od, rtol=1e-12, atol=0)


def test_fastcar_search_reproduces_reference_output(tmp_path):
    """query x database search (Predictor::close + similarity per pair) written like fastcar does; the file
    must equal what the reference's fastcar produced with --recover on the same model (tests/golden/fastcar_k5_u16.out)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "meshclust2_amd", "host", "msc_fastcar")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "meshclust2_amd", "host")])
    db, h = synth.families(41, 300, 1000, family=10, length_jitter=150)
    q, hq = synth.families(41, 40, 1000, family=10, length_jitter=150)
    q = [x[:len(x) - 7] for x in q]
    synth.write_fasta(str(tmp_path / "db.fa"), db, h)
    synth.write_fasta(str(tmp_path / "q.fa"), q, [x.replace(">seq", ">qry") for x in hq])
    golden = os.path.join(root, "tests", "golden")
    r = subprocess.run([exe, "db.fa", "--query", "q.fa", "--recover", os.path.join(golden, "weights_k5_u16.txt"), "--output", "fc_out"],
                       cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
    got = open(str(tmp_path / "fc_out0"), "rb").read()
    exp = open(os.path.join(golden, "fastcar_k5_u16.out"), "rb").read()
    assert got == exp, "fastcar output differs (%d vs %d bytes)" % (len(got), len(exp))


@pytest.mark.parametrize("mode", [1, 2])
def test_fastcar_follows_the_weights_file_mode(tmp_path, mode):
    """work() follows Predictor::get_mode (fastcar/FC_Runner.cpp:432,446-458): a classification-only file (`mode: 1`, what
    `meshclust2 --dump` / msc_train_class write) prints 100 for every close pair, a regression-only one (`mode: 2`) treats every
    pair of the length window as close. Fixtures: the reference's fastcar --recover on the same files."""
    import os
    import subprocess
    from golden_util import weights_with_mode
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "meshclust2_amd", "host", "msc_fastcar")
    db, h = synth.families(41, 300, 1000, family=10, length_jitter=150)
    q, hq = synth.families(41, 40, 1000, family=10, length_jitter=150)
    q = [x[:len(x) - 7] for x in q]
    hq = [x.replace(">seq", ">qry") for x in hq]
    if mode == 2:
        db, h, q, hq = db[:60], h[:60], q[:8], hq[:8]
    synth.write_fasta(str(tmp_path / "db.fa"), db, h)
    synth.write_fasta(str(tmp_path / "q.fa"), q, hq)
    open(str(tmp_path / "w.txt"), "w").write(weights_with_mode(weights_text("weights_k5_u16.txt"), mode))
    exp = open(os.path.join(root, "tests", "golden", "fastcar_k5_u16_mode%d.out" % mode), "rb").read()
    for qb in ("16", "1"):
        r = subprocess.run([exe, "db.fa", "--query", "q.fa", "--recover", "w.txt", "--output", "fc_out", "--query-block", qb],
                           cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
        got = open(str(tmp_path / "fc_out0"), "rb").read()
        assert got == exp, "fastcar output differs (%d vs %d bytes)" % (len(got), len(exp))


def test_cluster_driver_single_file_mode(tmp_path):
    """--single-file (each FASTA file is one sequence, records joined by 50 N): 48 three-record files,
    .clstr byte-identical to the reference CLI's."""
    import os
    import subprocess
    from golden_util import single_file_set
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "meshclust2_amd", "host", "msc_cluster")
    names = []
    for name, hdrs, recs in single_file_set():
        synth.write_fasta(str(tmp_path / name), recs, hdrs)
        names.append(name)
    golden = os.path.join(root, "tests", "golden")
    r = subprocess.run([exe] + names + ["--single-file", "--recover", os.path.join(golden, "weights_single_file_k5_u16.txt"), "--id", "0.85", "--kmer", "5",
                                        "--datatype", "16", "--output", "out.clstr"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
    assert open(str(tmp_path / "out.clstr"), "rb").read() == open(os.path.join(golden, "single_file.clstr"), "rb").read()


@pytest.mark.parametrize("dtype,k,length", [(32, 9, 1000), (8, 9, 1000), (16, 8, 3000), (32, 7, 400), (64, 10, 5000), (16, 11, 20000)])
def test_sparse_sets_equal_dense_sets(ctx, dtype, k, length):
    """The sparse layout (sorted (bin, value) lists + merge kernel) must reproduce the dense path: identical bins, scalars,
    integer statistics bit for bit, FP64 ones to 1e-9, and identical get_close / filter / merge / search decisions."""
    seqs, _ = synth.families(900 + k, 28, length, family=7, length_jitter=length // 10)
    seqs = list(seqs) + [seqs[0][: length // 2], b"ACGT" * 40 + b"N" * 30 + seqs[1][:300]]
    n = len(seqs)
    dense = api.HistogramSet(ctx, k, dtype, n + 1)
    sparse = api.HistogramSet(ctx, k, dtype, n + 1, sparse_entries=sum(len(s) for s in seqs) * 2 + 1000)
    dense.build(seqs)
    sparse.build(seqs)
    for i in range(n):
        assert sparse.info(i) == dense.info(i)
        assert sparse.entries(i) <= len(seqs[i])
        if k <= 9:
            assert np.array_equal(sparse.download(i), dense.download(i))
    cands = np.arange(n, dtype=np.uint32)[::-1].copy()
    for q in (0, 5, n - 1, n - 2):
        for order in (api.ORDER_CAND_FIRST, api.ORDER_QUERY_FIRST):
            a = api.pair_features_raw(ctx, sparse, cands, sparse, q, FAST_MASK, order)
            b = api.pair_features_raw(ctx, dense, cands, dense, q, FAST_MASK, order)
            for c, (name, _) in enumerate(FEATS):
                assert np.array_equal(a[:, c], b[:, c]), (name, q)          # the divergences too: one merge kernel scores them in both layouts
    wts = "weights_k9_u32.txt" if dtype == 32 else "weights_k5_u16_slow.txt"
    fs, fd = api.Feature.from_text(ctx, weights_text(wts), 0), api.Feature.from_text(ctx, weights_text(wts), 0)
    rs, rd = api.Feature.from_text(ctx, weights_text(wts), 1), api.Feature.from_text(ctx, weights_text(wts), 1)
    for cutoff in (0.9, 0.6):
        ts, td = api.Trainer(ctx, fs, cutoff), api.Trainer(ctx, fd, cutoff)
        for q in (0, 9, n - 1):
            w = np.array([c for c in range(n) if c != q], dtype=np.uint32)
            f1, bp1, bs1, im1 = ts.get_close(sparse, w, sparse, q)
            f2, bp2, bs2, im2 = td.get_close(dense, w, dense, q)
            assert np.array_equal(f1, f2) and (bp1, im1) == (bp2, im2) and bs1 == pytest.approx(bs2, rel=1e-12)
            assert np.array_equal(ts.filter(sparse, q, sparse, w), td.filter(dense, q, dense, w))
            assert ts.merge(sparse, cands[::-1].copy(), q, q + 1, min(n - 1, q + 6)) == td.merge(dense, cands[::-1].copy(), q, q + 1, min(n - 1, q + 6)) if q + 1 < n else True
        c1, s1 = api.Predictor(ctx, fs, rs).search(sparse, cands, sparse, 3)
        c2, s2 = api.Predictor(ctx, fd, rd).search(dense, cands, dense, 3)
        assert np.array_equal(c1, c2) and np.allclose(s1, s2, rtol=1e-9, atol=1e-12)
    # get_mean / closest on sparse members
    for mem in (np.arange(0, n, 2, dtype=np.uint32), np.arange(n, dtype=np.uint32), np.array([3], dtype=np.uint32), np.array([4, 4, 9], dtype=np.uint32)):
        p1, d1, _ = api.mean_nearest(ctx, sparse, mem)
        p2, d2, _ = api.mean_nearest(ctx, dense, mem)
        assert p1 == p2 and np.array_equal(d1, d2), mem
    # Center semantics on sparse slots
    sparse.clone_from(n, sparse, 0)
    sparse.assign_from(n, sparse, 5)
    dense.clone_from(n, dense, 0)
    dense.assign_from(n, dense, 5)
    assert sparse.info(n) == dense.info(n)
    a = api.pair_features_raw(ctx, sparse, [n], sparse, 3, FAST_MASK)
    b = api.pair_features_raw(ctx, dense, [n], dense, 3, FAST_MASK)
    assert np.allclose(a, b, rtol=1e-9, atol=1e-14)
    m = api.score_multi(ctx, fs, sparse, cands, sparse, [0, 3, 7])
    for i, q in enumerate((0, 3, 7)):
        assert np.array_equal(m["sum"][i], fd.compute(dense, cands, dense, q)["sum"]) or np.allclose(m["sum"][i], fd.compute(dense, cands, dense, q)["sum"], rtol=1e-9)


def test_sparse_k13_against_the_oracle(ctx, oracle):
    """k = 13 (67 M bins: the dense form of cfg4 cannot exist for a data set). A handful of 8-bit oracle
    histograms (64 MiB each on the host) pin the sparse path there."""
    k, dtype = 13, 8
    seqs, _ = synth.families(1313, 5, 6000, family=5)
    hs = api.HistogramSet(ctx, k, dtype, len(seqs), sparse_entries=40000)
    hs.build(seqs)
    oh = [oracle.hist(s, k, dtype) for s in seqs]
    for i in range(len(seqs)):
        inf = hs.info(i)
        assert (inf["mag"], inf["length"], inf["one_mers"]) == (oh[i].mag, oh[i].length, list(oh[i].one_mers))
        assert np.array_equal(hs.download(i), oh[i].array())
    cands = np.arange(len(seqs), dtype=np.uint32)
    raw = api.pair_features_raw(ctx, hs, cands, hs, 0, FAST_MASK)
    for c in range(len(seqs)):
        for col, (name, bit) in zip(raw[c], FEATS):
            exp = oracle.raw_feature(1 << bit, oh[c], oh[0])
            if name in EXACT and name != "kulczynski2":
                assert col == exp, (name, c)
            elif name == "pearson":
                # the reference adds 67 M FP64 terms one by one and is itself only good to ~1e-9 here; the GPU's closed form
                # over exact integer moments agrees with exact rational arithmetic to 1e-16 (checked offline)
                assert col == pytest.approx(exp, rel=1e-7), (name, c)
            else:
                assert col == pytest.approx(exp, rel=1e-9, abs=1e-13), (name, c)
    for h in oh:
        oracle.lib().orc_hist_free(h)


CFG4_WEIGHTS = weights_text("weights_cfg4_k13.txt")      # hand-written model (the reference cannot train at this size here)
_cfg4_sequences = cfg4_sequences


def _divergence_longdouble(name, a, b):
    """jefferey_divergence / jensen_shannon (predict/Feature.cpp:1231-1263,984-1009) of two oracle histograms in extended
    precision: the bins where either count differs from the pseudocount one by one, the (1, 1) term times its count"""
    p, q = a.array(), b.array()
    ld = np.longdouble
    mp_, mq_ = ld(int(a.mag)), ld(int(b.mag))
    idx = np.nonzero((p != 1) | (q != 1))[0]

    def term(pp, pq):
        if name == "jefferey_divergence":
            return (pp - pq) * np.log(pp / pq)
        avg = (pp + pq) / 2
        return (pp * np.log(pp / avg) + pq * np.log(pq / avg)) / 2
    total = np.sum(term(p[idx].astype(ld) / mp_, q[idx].astype(ld) / mq_)) + (p.size - idx.size) * term(ld(1) / mp_, ld(1) / mq_)
    return float(total)


@pytest.mark.parametrize("route", ["k_pair_sparse_mp", "k_pair_sparse"])
def test_cfg4_k13_u64_20kb_against_the_oracle(ctx, oracle, route):
    """BASELINE.json configs[3] at its stated parameters: k = 13, datatype = 64, 20 kb sequences, sparse layout (a dense k = 13
    uint64_t histogram is 512 MiB: SURVEY Q11; the oracle holds a handful of them on the host). Builders, all 11 statistics in
    both argument orders, one Trainer::get_close window, filter and the mean / closest step against oracle.hist(..., 13, 64).
    route k_pair_sparse_mp: counts < 2^16, the merge-path kernel; route k_pair_sparse: one sequence with a 70 000-base
    homopolymer run forces the 64-bit lane-per-sub-range kernel (clutil/Loader.cpp:42-86, predict/Feature.cpp:984-1009,1231-1263)."""
    k, dtype = 13, 64
    seqs, mono = _cfg4_sequences()
    if route == "k_pair_sparse":
        seqs = seqs[:2] + [mono] + seqs[3:4]
    n = len(seqs)
    hs = api.HistogramSet(ctx, k, dtype, n + 1, sparse_entries=sum(len(s) for s in seqs) + 4096)
    hs.build(seqs)
    oh = [oracle.hist(s, k, dtype) for s in seqs]
    for i in range(n):
        inf = hs.info(i)
        assert (inf["mag"], inf["length"], inf["one_mers"], inf["overflow"]) == (oh[i].mag, oh[i].length, list(oh[i].one_mers), oh[i].overflow), i
        # clutil/Loader.cpp:158-171 adds 67 M FP64 squares one by one: its own rounding error is ~1e-10 relative at this size; the
        # GPU value is a closed form over exact integer moments (the k <= 11 tests hold 1e-10)
        assert inf["stddev"] == pytest.approx(oh[i].stddev, rel=1e-8)
        assert np.array_equal(hs.download(i), oh[i].array()), i
    if route == "k_pair_sparse":
        assert hs.info(2)["max_count"] >= 1 << 16
    cands = np.arange(n, dtype=np.uint32)
    for q in ((0, n - 1) if route == "k_pair_sparse_mp" else (2, 0)):
        for order in (api.ORDER_CAND_FIRST, api.ORDER_QUERY_FIRST):
            raw = api.pair_features_raw(ctx, hs, cands, hs, q, FAST_MASK, order)
            assert ctx.last_kernel_info()[0] == route
            for c in range(n):
                a, b = (oh[c], oh[q]) if order == api.ORDER_CAND_FIRST else (oh[q], oh[c])
                for col, (name, bit) in zip(raw[c], FEATS):
                    exp = oracle.raw_feature(1 << bit, a, b)
                    if name in EXACT and name != "kulczynski2":
                        assert col == exp, (name, c, q, order)
                    elif name == "pearson":
                        # 67 M FP64 terms added one by one: the reference's own value is only good to ~1e-9 here (see the k13/u8 test)
                        assert col == pytest.approx(exp, rel=1e-7), (name, c, q)
                    elif name in ("jefferey_divergence", "jensen_shannon"):
                        # the same for the two divergences: the reference adds 67 M terms (almost all the tiny (1, 1) term) one by
                        # one and its sum is 1e-9 off the exact value; the GPU adds the few thousand other terms and multiplies the
                        # (1, 1) term by its count. Held to the reference at 1e-7 (the north-star bar is 1e-5) and to an
                        # extended-precision evaluation of the same formula at 1e-10.
                        assert col == pytest.approx(exp, rel=1e-7, abs=1e-13), (name, c, q, order)
                        # (1e-10: the two logs of a jensen_shannon term cancel to second order, so FP64 terms carry ~1e-12 each)
                        assert col == pytest.approx(_divergence_longdouble(name, a, b), rel=1e-10, abs=1e-18), (name, c, q, order)
                    else:
                        assert col == pytest.approx(exp, rel=1e-9, abs=1e-13), (name, c, q, order)
    feat = api.Feature.from_text(ctx, CFG4_WEIGHTS, 0)
    pred = oracle.predictor(CFG4_WEIGHTS)
    for cutoff in (0.9, 0.6):
        trn = api.Trainer(ctx, feat, cutoff)
        for q in range(n):
            w = np.array([c for c in range(n) if c != q], dtype=np.uint32)
            flags, bp, bs, im = trn.get_close(hs, w, hs, q)
            of, obp, obs, oim = oracle.get_close(pred, cutoff, oh[q], [oh[c] for c in w])
            assert np.array_equal(flags, of) and (bp, im) == (obp, oim), (cutoff, q)
            assert bs == pytest.approx(obs, rel=1e-7), (cutoff, q)          # jensen_shannon is in the model: see the 1e-9 note above
            assert np.array_equal(trn.filter(hs, q, hs, w), oracle.filter_(pred, cutoff, oh[q], [oh[c] for c in w])), (cutoff, q)
    r = feat.compute(hs, cands, hs, 1)
    for c in range(n):
        s_, _, wsum = oracle.score(pred.cls, oh[c], oh[1])
        assert np.allclose(r["singles"][c], s_, rtol=1e-7, atol=1e-10) and r["sum"][c] == pytest.approx(wsum, rel=1e-7, abs=1e-9), c
    if route == "k_pair_sparse_mp":
        mem = np.array([0, 1, 2, 4], dtype=np.uint32)
        pos, d, _ = api.mean_nearest(ctx, hs, mem)
        _, od, opos = oracle.mean_nearest([oh[i] for i in mem])
        assert pos == opos and np.allclose(d, od, rtol=1e-12, atol=0)
    for h in oh:
        oracle.lib().orc_hist_free(h)


def test_cfg4_k13_u64_values_printed_by_the_reference(ctx):
    """BASELINE cfg4's parameters against the REFERENCE's own numbers, not only its restatement: tests/golden/vectors_k13_u64.npz was
    printed by /root/reference code (gen_golden.make_vectors_k13: Loader::get_point at k = 13 / uint64_t -- 512 MiB tables --, the 11
    statistics of predict/Feature.cpp:682-1518 for every ordered pair of six 17-90 kb sequences, Trainer::get_close / filter under
    weights_cfg4_k13.txt, get_mean + distance_d). Counts below 2^16 take the merge-path kernel, the homopolymer sequence (one bin of
    69 988) the 64-bit lane-per-sub-range kernel. Integer statistics: equal; FP64 sums of 67 M terms: 1e-7 (the reference's own rounding, see
    test_cfg4_k13_u64_20kb_against_the_oracle), decisions and arg-max: equal."""
    v = load_vectors("vectors_k13_u64.npz")
    k, dtype, n = int(v["k"]), int(v["dtype"]), int(v["n"])
    seqs = [bytes(s_) for s_ in v["seqs"]]
    hs = api.HistogramSet(ctx, k, dtype, n + 1, sparse_entries=sum(len(s_) for s_ in seqs) + 4096)
    hs.build(seqs)
    for i in range(n):
        inf = hs.info(i)
        assert (inf["mag"], inf["length"], inf["one_mers"]) == (int(v["mag"][i]), int(v["length"][i]), v["one_mers"][i].tolist()), i
        assert inf["stddev"] == pytest.approx(float(v["stddev"][i]), rel=1e-8)
        assert hs.entries(i) == v["bins_idx_%d" % i].size
    cands = np.arange(n, dtype=np.uint32)
    for q in range(n):
        raw = api.pair_features_raw(ctx, hs, cands, hs, q, FAST_MASK | api.FEAT["jefferey_divergence"] | api.FEAT["jensen_shannon"], api.ORDER_CAND_FIRST)
        for c in range(n):
            for col, (f, (name, bit)) in zip(raw[c], enumerate(FEATS)):
                exp = v["raw"][c, q, f]          # f(first = candidate, second = query)
                if name in EXACT and name != "kulczynski2":
                    assert col == exp, (name, c, q)
                elif name in ("pearson", "jefferey_divergence", "jensen_shannon"):
                    assert col == pytest.approx(exp, rel=1e-7, abs=1e-13), (name, c, q)
                else:
                    assert col == pytest.approx(exp, rel=1e-9, abs=1e-13), (name, c, q)
    feat = api.Feature.from_text(ctx, weights_text("weights_cfg4_k13.txt"), 0)
    for ci, cutoff in enumerate(v["cutoffs"]):
        trn = api.Trainer(ctx, feat, float(cutoff))
        for q in range(n):
            w = np.array([c for c in range(n) if c != q], dtype=np.uint32)
            flags, bp, bs, im = trn.get_close(hs, w, hs, q)
            gbp, gbs, gim = v["get_close_best_%d" % ci][q]
            assert np.array_equal(flags, v["get_close_flags_%d" % ci][q]) and (bp, im) == (int(gbp), bool(gim)), (cutoff, q)
            assert bs == pytest.approx(gbs, rel=1e-7), (cutoff, q)
            assert np.array_equal(trn.filter(hs, q, hs, w), v["filter_%d" % ci][q]), (cutoff, q)
    r = feat.compute(hs, cands, hs, 1)
    assert np.allclose(r["sum"], v["sums"][:, 1], rtol=1e-7, atol=1e-9)
    pos, d, _ = api.mean_nearest(ctx, hs, v["mean_members"].astype(np.uint32))
    # (distance_d = 10000 (1 - frac^2) with frac = 0.99998: 11 digits survive the subtraction)
    assert pos == int(v["mean_nearest"]) and np.allclose(d, v["mean_dist"], rtol=1e-9, atol=0)


@pytest.mark.parametrize("layout", ["dense", "sparse"])
def test_cluster_driver_k8_dense_and_sparse_layouts(tmp_path, layout):
    """k = 8 / 16-bit bins, 640 sequences: the driver writes the reference's .clstr bytes with the dense layout AND with the
    sparse one (--sparse: sorted (bin, value) lists, merge kernel, sparse mean/closest, sparse centre store)."""
    import os
    import subprocess
    from golden_util import k8_set
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "meshclust2_amd", "host", "msc_cluster")
    seqs, hdrs = k8_set()
    fa = str(tmp_path / "k8.fa")
    synth.write_fasta(fa, seqs, hdrs)
    golden = os.path.join(root, "tests", "golden")
    args = [exe, fa, "--recover", os.path.join(golden, "weights_k8_u16.txt"), "--id", "0.85", "--kmer", "8", "--datatype", "16", "--output", "out.clstr"]
    if layout == "sparse":
        args.append("--sparse")
    r = subprocess.run(args, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
    assert open(str(tmp_path / "out.clstr"), "rb").read() == open(os.path.join(golden, "k8.clstr"), "rb").read()


def test_dense_one_by_m_passes_stream_the_bins_when_told_to(tmp_path):
    """Since r04 a dense set's 1 x M passes merge the lists of its sparse mirror (histograms of 64 KiB and more). The streaming kernel
    over the bins (k_pair_tiles: SURVEY 8(d)'s 1 x M shape) stays behind msc_set_mirror_pass(0) / MSC_NO_MIRROR_1XM: the fixture and
    oracle tests of this file once more with the switch set, in a process of their own, and both forms side by side here."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    ctx = api.Context(0)
    seqs, _ = synth.families(4242, 60, 1000, family=6, length_jitter=90)
    hs = api.HistogramSet(ctx, 9, 32, len(seqs))
    hs.build(seqs)
    feat = api.Feature.from_text(ctx, weights_text("weights_k9_u32.txt"), 0)
    a = api.pair_features_raw(ctx, hs, None, hs, 3, FAST_MASK, m=len(seqs))
    ka = ctx.last_kernel_info()[0]
    sa = feat.compute(hs, None, hs, 3, m=len(seqs))
    ctx.set_mirror_pass(False)
    b = api.pair_features_raw(ctx, hs, None, hs, 3, FAST_MASK, m=len(seqs))
    kb = ctx.last_kernel_info()[0]
    sb = feat.compute(hs, None, hs, 3, m=len(seqs))
    assert ka.startswith("k_pair_sparse") and kb == "k_pair_tiles", (ka, kb)
    assert np.array_equal(a, b) and np.array_equal(sa["sum"], sb["sum"]) and np.array_equal(sa["csum"], sb["csum"])
    ctx.close()
    env = dict(os.environ, MSC_NO_MIRROR_1XM="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_parity.py"), "-q", "-m", "gpu", "-x", "-k",
                        "golden_vectors or against_oracle_seeded or appendix_d"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-3000:]


def test_full_size_cfg2_properties(oracle):
    """BASELINE.json configs[1] at FULL size (100 000 x 1 kb, k = 9, datatype 32: 98 GiB of histograms + the digest mirror),
    checked through size-independent properties: sampled histograms against the oracle; the Q x M pass on the matrix cores
    (k_pair_gemm_fp4_dma: 768 whole tiles of 128 candidates + the 14 left over cut into short pieces that add their sums) against the
    independent 1 x M kernel bit for bit over all 100 000 candidates; the close flags of a flags-only call (f32 screen + FP64 for the
    undecided) against those of the FP64 evaluation; symmetry of the statistics; the self pair; the family structure of the synthetic
    set (20 relatives per template, ~96 % identical)."""
    ctx = api.Context(0)
    n, k, dtype, fam = 100000, 9, 32, 20
    seed = 20260002
    hs = api.HistogramSet(ctx, k, dtype, n)
    keep = {}
    for done in range(0, n, 20000):
        codes = []
        for t in range(done // fam, (done + 20000) // fam):
            tmpl = synth.template(seed, t, 1000)
            codes += [synth.member(seed, t, j, tmpl) for j in range(fam)]
        for i in (0, 7777, 19999):
            keep[done + i] = codes[i]
        b = synth.pack_batch(codes)
        hs.build_packed(done, 20000, b["packed"], b["n_bases"], b["seg_seq"], b["seg_start"], b["seg_end"], b["eff_len"], b["one_mers"])
    # sampled histograms == oracle (codes 0..3 -> ACGT)
    for slot, code in keep.items():
        seq = synth.to_ascii(code)
        oh = oracle.hist(seq, k, dtype)
        assert np.array_equal(hs.download(slot), oh.array()), slot
        inf = hs.info(slot)
        assert inf["sum"] == len(seq) - k + 1 + 4 ** k and inf["mag"] == oh.mag and inf["length"] == len(seq)
    feat = api.Feature.from_text(ctx, weights_text("weights_k9_u32.txt"), 0)
    qs = (np.arange(16, dtype=np.uint32) * 6151 + 3) % n
    fast_mask = sum(1 << b for name, b in FEATS if name not in ("jefferey_divergence", "jensen_shannon"))
    multi = api.score_multi(ctx, feat, hs, None, hs, qs, m=n, feat_mask=(1 << 2) | (1 << 13))
    assert ctx.last_kernel_info()[0].startswith("k_pair_gemm_fp4_dma<32 query rows"), ctx.last_kernel_info()          # cfg2 itself: the matrix-core route, its default kernel
    only = api.score_multi(ctx, feat, hs, None, hs, qs, m=n, want=("close", "counts"))
    assert np.array_equal(only["close"], multi["close"]) and np.array_equal(only["counts"], multi["close"].sum(axis=1, dtype=np.uint64))
    # independent kernel, same answers (every candidate, three of the queries)
    for i in (0, 5, 15):
        single = feat.compute(hs, None, hs, int(qs[i]), m=n)
        assert ctx.last_kernel_info()[0] in ("k_pair_tiles", "k_pair_sparse_wl", "k_pair_sparse_mp", "k_pair_ranks_1xm"), ctx.last_kernel_info()          # (r04: the dense set's mirror pass)
        assert np.array_equal(multi["sum"][i], single["sum"]) and np.array_equal(multi["csum"][i], single["csum"])
        assert int(multi["close"][i].sum()) == int((np.round(single["csum"]) > 0).sum())
    # first-hand at full size: the pass's weighted sums, close flags and integer statistics for the kept candidates x all 16 queries
    # next to the CPU oracle (predict/Feature.cpp:156-171, cluster/Trainer.cpp:49-52) -- 15 x 16 pairs out of the 1.6e6 scored
    pred = oracle.predictor(weights_text("weights_k9_u32.txt"))
    q_h = [oracle.hist(synth.to_ascii(synth.member(seed, int(q) // fam, int(q) % fam, synth.template(seed, int(q) // fam, 1000))), k, dtype) for q in qs]
    for slot, code in keep.items():
        c_h = oracle.hist(synth.to_ascii(code), k, dtype)
        for i in range(16):
            _, _, w = oracle.score(pred.cls, c_h, q_h[i])
            assert multi["sum"][i][slot] == pytest.approx(w, rel=1e-8, abs=1e-10), (slot, i)
            assert multi["close"][i][slot] == (1 if round(1.0 / (1.0 + np.exp(-w))) > 0 else 0), (slot, i)
            assert multi["raw"][i][slot][0] == oracle.raw_feature(1 << 2, c_h, q_h[i]), (slot, i)          # manhattan, bitwise
            assert multi["raw"][i][slot][1] == oracle.raw_feature(1 << 13, c_h, q_h[i]), (slot, i)         # intersection, bitwise
        oracle.lib().orc_hist_free(c_h)
    for h_ in q_h:
        oracle.lib().orc_hist_free(h_)
    # symmetry of the symmetric statistics: stat(query a, candidate b) == stat(query b, candidate a). (The model score itself is
    # NOT symmetric for uint32_t bins: the reference's simratio wraps `p - q` before widening, SURVEY Q3, reproduced here.)
    for i in range(16):
        for j in range(i + 1, 16):
            assert np.array_equal(multi["raw"][i][qs[j]], multi["raw"][j][qs[i]])
    for i, q in enumerate(qs):
        # the self pair: manhattan 0, intersection 1 (whether the trained model calls it close is the model's business)
        assert multi["raw"][i][q][0] == 0.0 and multi["raw"][i][q][1] == 1.0
        # every relative of the query's template (20 per family, ~96 % identical) shares more k-mers with it than any stranger
        f0 = (int(q) // fam) * fam
        inter = multi["raw"][i][:, 1]
        strangers = np.concatenate([inter[:f0], inter[f0 + fam:]])
        assert inter[f0:f0 + fam].min() > strangers.max(), (i, inter[f0:f0 + fam].min(), strangers.max())
    assert fast_mask
    ctx.close()


def test_runs_of_more_than_a_million_bases_through_the_build(ctx, oracle):
    """msc_hist_build on sequences whose runs of unambiguous bases pass 1 000 000 (Chromosome::makeSegmentList under help(1000000),
    nonltr/Chromosome.cpp:355-385,115-128: fragments of 1 000 000, the last one longer; k-mers across a cut are not counted): segments
    from msc_encode, histograms, magnitude, length and 1-mers next to the reference's own (tests/golden/long_fragments.npz), dense and
    sparse, and the 9-mer histograms next to the oracle (the k-mer count drops by k - 1 per cut)."""
    import os
    from golden_util import long_fragment_set
    v = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "long_fragments.npz"))
    seqs = long_fragment_set()
    for i, s in enumerate(seqs):
        _, segs, eff = api.encode(s)
        assert [list(x) for x in segs] == v["segs_%d" % i].tolist() and eff == int(v["eff_%d" % i])
    for k, dt in ((5, 32), (7, 16)):
        hs = api.HistogramSet(ctx, k, dt, len(seqs))
        hs.build(seqs)
        for i in range(len(seqs)):
            assert np.array_equal(hs.download(i), v["bins_k%d_%d" % (k, i)]), (k, i)
            meta, inf = v["meta_k%d_%d" % (k, i)], hs.info(i)
            assert (inf["mag"], inf["length"]) == (int(meta[0]), int(meta[1])) and inf["one_mers"] == meta[2:].tolist()
            assert inf["stddev"] == pytest.approx(float(v["stddev_k%d_%d" % (k, i)]), rel=1e-12)
    for sparse in (False, True):          # (the list form exists from 64 KiB histograms up)
        hs = api.HistogramSet(ctx, 9, 32, 2, sparse_entries=2 * 4 ** 9 if sparse else 0)
        hs.build(seqs[1:])
        for i, s in enumerate(seqs[1:]):
            oh = oracle.hist(s, 9, 32)
            assert np.array_equal(hs.download(i), oh.array()), (i, sparse)
            _, segs, _ = api.encode(s)
            assert hs.info(i)["sum"] - 4 ** 9 == sum(e - b + 1 - 8 for b, e in segs if e - b + 1 >= 9)
            oracle.lib().orc_hist_free(oh)


@pytest.mark.parametrize("dtype,k,wts,sparse", [(16, 5, "weights_k5_u16.txt", False), (32, 9, "weights_k9_u32.txt", False), (16, 5, "weights_k5_u16_slow.txt", False),
                                                 (8, 3, "weights_k5_u16.txt", False), (32, 9, "weights_k9_u32.txt", True), (16, 8, "weights_k8_u16.txt", True),
                                                 (8, 9, "weights_k9_u8.txt", True), (64, 10, "weights_k5_u16.txt", True),
                                                 # `--feat slow` models where the list form exists: the pair-list divergence pass (r02)
                                                 (16, 9, "weights_cfg5_u16_k9.txt", False), (16, 9, "weights_cfg5_u16_k9.txt", True),
                                                 (8, 9, "weights_cfg5_k9.txt", False), (8, 9, "weights_cfg5_k9.txt", True),
                                                 # k >= 11: the sweeps of the batched sparse mean visit touched 64-byte lines only (r02)
                                                 (8, 11, "weights_k9_u8.txt", True), (64, 13, "weights_k5_u16.txt", True)])
def test_batched_update_and_merge_equal_the_per_centre_calls(ctx, dtype, k, wts, sparse):
    """msc_update_centres / msc_merge_all (one launch per stage for all centres of a round) == msc_filter + msc_mean_nearest /
    msc_merge centre by centre: ragged and empty lists, lists nothing survives, the divergence statistics (a pair-list pass of the
    chunked merge kernel over the lists / the mirrors' lists; centre by centre where no list form exists), padded tiny histograms; on sparse sets too (r02: pair-list merge-path kernel + the scatter / sweep of the rounded means with a centre
    dimension), where the per-centre calls are the single-query kernels."""
    rng = np.random.default_rng(11 * k + dtype)
    seqs, _ = synth.families(4100 + k, 120, 600 if k > 3 else 80, family=6)
    seqs = [s[: len(s) - int(rng.integers(0, len(s) // 3))] for s in seqs]      # mixed lengths: the length window matters
    n = len(seqs)
    arena = (sum(len(s_) for s_ in seqs) * 3 + 4096) if sparse else 0
    pts = api.HistogramSet(ctx, k, dtype, n, sparse_entries=arena)
    pts.build(seqs)
    feat = api.Feature.from_text(ctx, weights_text(wts), 0)
    trn = api.Trainer(ctx, feat, 0.9)
    nc = 37
    cen = api.HistogramSet(ctx, k, dtype, nc, sparse_entries=arena)
    owners = rng.permutation(n)[:nc]
    for c, o in enumerate(owners):
        cen.clone_from(c, pts, int(o))
    cslots = rng.permutation(nc).astype(np.uint32)
    lists = [rng.permutation(n)[: int(rng.integers(0, 40))].astype(np.uint32) for _ in range(nc)]
    lists[3] = np.zeros(0, dtype=np.uint32)
    nearest, kept = trn.update_centres(cen, cslots, pts, lists)
    for c in range(nc):
        keep = trn.filter(cen, int(cslots[c]), pts, lists[c]) if len(lists[c]) else np.zeros(0, dtype=np.uint8)
        idx = np.flatnonzero(keep)
        assert kept[c] == idx.size, c
        if idx.size == 0:
            assert nearest[c] == -1, c
        else:
            pos, _, _ = trn.closest(pts, lists[c][idx])
            assert nearest[c] == idx[pos], c
    for delta in (0, 1, 5):
        best = trn.merge_all(cen, cslots, delta)
        for i in range(nc):
            exp = trn.merge(cen, cslots, i, i + 1, min(nc - 1, i + delta))
            assert best[i] == exp, (delta, i)
        # ... and asked about some centres only (msc_merge_some: the driver's merge round once the clusters have settled), in any order
        which = np.array([i for i in range(nc) if i % 3 != 1][::-1], dtype=np.uint64)
        assert list(trn.merge_some(cen, cslots, delta, which)) == [best[int(i)] for i in which]
        assert list(trn.merge_some(cen, cslots, delta, np.array([], dtype=np.uint64))) == []


def test_cluster_driver_batched_update_equals_serial(tmp_path):
    """The driver's batched update stage (msc_update_centres + msc_merge_all per round) writes the same .clstr bytes as the
    centre-by-centre order (--serial-update) on a 3000-sequence set that ends in ~1900 clusters."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seqs, headers = synth.families(777, 3000, 1000)
    fa = str(tmp_path / "in.fa")
    synth.write_fasta(fa, seqs, headers)
    outs = []
    for extra in ([], ["--serial-update"]):
        out = str(tmp_path / ("o%d.clstr" % len(outs)))
        r = subprocess.run([os.path.join(root, "meshclust2_amd", "host", "msc_cluster"), fa, "--recover", os.path.join(root, "tests", "golden", "weights_k5_u16.txt"),
                            "--id", "0.9", "--kmer", "5", "--datatype", "16", "--output", out] + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] and outs[0].count(b">Cluster") > 1000


@pytest.mark.parametrize("fixture", ["train_k5_u16.json", "train_k7_u8_slow.json", "train_k9_u32.json"])
def test_training_selects_the_reference_model(ctx, fixture):
    """msc_train_class (GPU feature table + host best-first selection + normal equations) on the labelled pairs of
    a fixture picks the SAME combos and single features, in the same order, as the reference's BestFirstSelector::train_class did,
    with weights within 1e-6 relative, bounds within 1e-9 and identical accuracies; the text it writes loads as a model."""
    import json
    import os
    from golden_util import GOLDEN, parse_class_block, training_set
    fx = json.load(open(os.path.join(GOLDEN, fixture)))
    seqs, pairs = training_set(fx["seed"], fx["n_templates"], fx["per_template"], fx["length"])
    assert [[a, b] for a, b, _ in pairs] == [[a, b] for a, b, _ in fx["pairs"]]
    pts = api.HistogramSet(ctx, fx["k"], fx["dtype"], len(seqs))
    pts.build(seqs)
    text, atr, ate = api.train_class(ctx, pts, [p[0] for p in fx["pairs"]], [p[1] for p in fx["pairs"]], [p[2] for p in fx["pairs"]], fx["n_train"],
                                     fx["feat_flags"], fx["min_feat"], fx["max_feat"], fx["id"])
    w0, combos, singles = parse_class_block(text)
    ew0, ecombos, esingles = parse_class_block(fx["block"])
    assert [(c, f) for c, f, _ in combos] == [(c, f) for c, f, _ in ecombos]
    assert [f for f, _, _ in singles] == [f for f, _, _ in esingles]
    assert w0 == pytest.approx(ew0, rel=1e-6)
    for (_, _, w), (_, _, ew) in zip(combos, ecombos):
        assert w == pytest.approx(ew, rel=1e-6)
    for (_, lo, hi), (_, elo, ehi) in zip(singles, esingles):
        assert lo == pytest.approx(elo, rel=1e-9, abs=1e-12) and hi == pytest.approx(ehi, rel=1e-9, abs=1e-12)
    assert atr == pytest.approx(fx["train_acc"], abs=1e-9) and ate == pytest.approx(fx["test_acc"], abs=1e-9)
    feat = api.Feature.from_text(ctx, text, 0)          # a complete weights file: header + class block
    r = feat.compute(pts, np.arange(4, dtype=np.uint32), pts, 0)
    assert np.all(np.isfinite(r["sum"]))


@pytest.mark.parametrize("fixture", ["train_regr_k5_u16.json", "train_regr_k7_u8_slow.json"])
def test_regression_training_selects_the_reference_model(ctx, fixture):
    """msc_train_regr (Predictor<T>::train_regr -> GreedySelector<T>::train_regression, predict/Predictor.cpp:977-985, predict/GreedySelector.cpp:11-76)
    on the labelled pairs of a fixture: the SAME combos in the same order of acceptance, the same single features and bounds, weights
    within 1e-6 and the same mean errors as the reference's own objects gave (oracle/ref_harness.cpp follows the body of its selector,
    which cannot be run to its end); the text loads as the regression block of a model and predicts inside [0, 1]."""
    import json
    import os
    from golden_util import GOLDEN, parse_class_block, training_set
    fx = json.load(open(os.path.join(GOLDEN, fixture)))
    seqs, pairs = training_set(fx["seed"], fx["n_templates"], fx["per_template"], fx["length"])
    pairs = [p for p in pairs if p[2] > fx["id"]]
    assert [[a, b] for a, b, _ in pairs] == [[a, b] for a, b, _ in fx["pairs"]]
    pts = api.HistogramSet(ctx, fx["k"], fx["dtype"], len(seqs))
    pts.build(seqs)
    text, etr, ete = api.train_regr(ctx, pts, [p[0] for p in fx["pairs"]], [p[1] for p in fx["pairs"]], [p[2] for p in fx["pairs"]], fx["n_train"], fx["feat_flags"],
                                    fx["max_feat"], fx["id"])
    w0, combos, singles = parse_class_block(text)
    ew0, ecombos, esingles = parse_class_block(fx["block"])
    assert [(c, f) for c, f, _ in combos] == [(c, f) for c, f, _ in ecombos]
    assert [f for f, _, _ in singles] == [f for f, _, _ in esingles]
    # integer-derived statistics only (k5_u16): the same feature table to the last bit, weights at 1e-6. With jefferey / jensen_shannon in
    # the table (k7_u8_slow) its entries differ from the reference's by the 1e-13 of another summation order, which the normal
    # equations of an identity fit on near-collinear combos amplify: 6.6e-6 on one weight (the mean errors follow).
    wtol = 1e-6 if "slow" not in fixture else 5e-5
    assert w0 == pytest.approx(ew0, rel=wtol)
    for (_, _, w), (_, _, ew) in zip(combos, ecombos):
        assert w == pytest.approx(ew, rel=wtol)
    for (_, lo, hi), (_, elo, ehi) in zip(singles, esingles):
        assert lo == pytest.approx(elo, rel=1e-9, abs=1e-12) and hi == pytest.approx(ehi, rel=1e-9, abs=1e-12)
    assert etr == pytest.approx(fx["train_err"], rel=wtol) and ete == pytest.approx(fx["test_err"], rel=wtol)
    assert "mode: 2" in text
    pred = api.Predictor.from_text(ctx, text)          # regression only: every entry close, similarity = clamp(prediction)
    close, sim = pred.search(pts, np.arange(6, dtype=np.uint32), pts, 0)
    assert close.all() and np.all((sim >= 0) & (sim <= 1)) and sim[0] > 0.9


def test_fastcar_with_a_reference_trained_regression_block(tmp_path):
    """SURVEY 8(f4) with a `mode: 3` weights file whose two blocks were BOTH printed by the reference (weights_k5_u16_regr.txt: the class
    block the reference CLI trained + the regression block of train_regr_k5_u16.json; r02's files carry a hand-assembled one):
    msc_fastcar writes the reference fastcar's own output byte for byte (fastcar/FC_Runner.cpp:426-471)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "meshclust2_amd", "host", "msc_fastcar")
    db, h = synth.families(41, 300, 1000, family=10, length_jitter=150)
    q, hq = synth.families(41, 40, 1000, family=10, length_jitter=150)
    q = [x[:len(x) - 7] for x in q]
    synth.write_fasta(str(tmp_path / "db.fa"), db, h)
    synth.write_fasta(str(tmp_path / "q.fa"), q, [x.replace(">seq", ">qry") for x in hq])
    golden = os.path.join(root, "tests", "golden")
    r = subprocess.run([exe, "db.fa", "--query", "q.fa", "--recover", os.path.join(golden, "weights_k5_u16_regr.txt"), "--output", "fc_out"],
                       cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
    got = open(str(tmp_path / "fc_out0"), "rb").read()
    exp = open(os.path.join(golden, "fastcar_k5_u16_regr.out"), "rb").read()
    assert got == exp, "fastcar output differs (%d vs %d bytes)" % (len(got), len(exp))


def test_fastcar_query_blocks_agree(tmp_path):
    """msc_fastcar scores queries in blocks of 16 length-neighbours through msc_score_multi (one pass over the union of their
    length windows); the output is the same bytes as with one query per pass, and as with an odd block size."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seqs, headers = synth.families(91, 3000, 1000, length_jitter=150)
    db, q = str(tmp_path / "db.fa"), str(tmp_path / "q.fa")
    synth.write_fasta(db, seqs, headers)
    synth.write_fasta(q, seqs[5:905:3], headers[5:905:3])
    outs = []
    for qb in (1, 7, 16):
        prefix = str(tmp_path / ("fc%d_" % qb))
        r = subprocess.run([os.path.join(root, "meshclust2_amd", "host", "msc_fastcar"), db, "--query", q, "--recover", os.path.join(root, "tests", "golden", "weights_k5_u16.txt"),
                            "--output", prefix, "--query-block", str(qb)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
        outs.append(open(prefix + "0", "rb").read())
    assert outs[0] == outs[1] == outs[2] and outs[0].count(b"\n") > 1000


def test_multi_query_pass_with_queries_from_another_set(ctx):
    """Q x M with the queries in a DIFFERENT set from the candidates (fastcar's query / database chunks): both sets get their own
    digest mirror; uploading into the query set afterwards refreshes just those slots."""
    k, dtype = 8, 32
    seqs, _ = synth.families(611, 90, 1000, family=9)
    db = api.HistogramSet(ctx, k, dtype, 70)
    db.build(seqs[:70])
    qs_set = api.HistogramSet(ctx, k, dtype, 20)
    qs_set.build(seqs[70:])
    feat = api.Feature.from_text(ctx, weights_text("weights_k9_u32.txt"), 0)
    q_slots = np.arange(20, dtype=np.uint32)[::-1].copy()
    mask = FAST_MASK & ~((1 << 7) | (1 << 29))
    for rnd in range(2):
        multi = api.score_multi(ctx, feat, db, None, qs_set, q_slots, m=70, feat_mask=mask)
        assert ctx.last_kernel_info()[0].startswith(("k_pair_digest_multi", "k_pair_gemm_fp4_dma<"))
        for i, q in enumerate(q_slots):
            raw = api.pair_features_raw(ctx, db, None, qs_set, int(q), mask, m=70)
            single = feat.compute(db, None, qs_set, int(q), m=70)
            assert np.array_equal(multi["raw"][i], raw) and np.array_equal(multi["sum"][i], single["sum"]), (rnd, i)
        qs_set.upload(4, db.download(11), 1000)         # overwrite one query slot with a database histogram
        qs_set.clone_from(9, db, 30)


@pytest.mark.parametrize("dtype,k,nq", [(32, 8, 24), (32, 8, 7), (16, 9, 40), (8, 8, 64)])
def test_multi_query_emd_by_ranks_mixed_lengths(ctx, dtype, k, nq):
    """The earth mover's distance of the Q x M pass from sorted k-mer ranks (msc_emd_ranks.hip): lists of very different lengths,
    several rounds of 1 024 ranks, repeats (counts > 2), queries from a set with another pitch -- the same integers, hence the same
    statistics and scores, as the per-bin prefix walk of the 1 x M pass."""
    rng = np.random.default_rng(900 + k + nq)
    seqs = []
    for i in range(90):
        n = int(rng.integers(200, 5200))
        s = "".join("ACGT"[b] for b in rng.integers(0, 4, n))
        if i % 5 == 0:
            s = s[: n // 2] + "ACGTTGCA" * (n // 40) + s[n // 2:]          # a repeat: some bins counted many times
        seqs.append(s)
    db = api.HistogramSet(ctx, k, dtype, 70)
    db.build(seqs[:70])
    qset = api.HistogramSet(ctx, k, dtype, 64)
    short = [s[:700] for s in seqs]
    qset.build((seqs[70:] + short)[:64])          # the first 20 long, the rest short
    feat = api.Feature.from_text(ctx, weights_text("weights_k9_u32.txt").replace("k: 9", "k: %d" % k).replace("uint32_t", "uint%d_t" % dtype), 0)
    mask = FAST_MASK & ~((1 << 7) | (1 << 29))
    qs = np.arange(nq, dtype=np.uint32)[::-1].copy()
    multi = api.score_multi(ctx, feat, db, None, qset, qs, m=70, feat_mask=mask)
    name = ctx.last_kernel_info()[0]
    assert "emd by ranks" in name, name
    for i, q in enumerate(qs):
        raw = api.pair_features_raw(ctx, db, None, qset, int(q), mask, m=70)
        single = feat.compute(db, None, qset, int(q), m=70)
        assert np.array_equal(multi["raw"][i], raw), (i, name)
        assert np.array_equal(multi["sum"][i], single["sum"]) and np.array_equal(multi["csum"][i], single["csum"]), i
    # a longer histogram arrives in the candidate set: the mirror is laid out again with the new pitch
    long_seq = "".join("ACGT"[b] for b in rng.integers(0, 4, 9000))
    db.build([long_seq], first_slot=5)
    multi = api.score_multi(ctx, feat, db, None, qset, qs, m=70, feat_mask=mask)
    for i, q in enumerate(qs[:6]):
        raw = api.pair_features_raw(ctx, db, None, qset, int(q), mask, m=70)
        assert np.array_equal(multi["raw"][i], raw), i


@pytest.mark.parametrize("dtype,k,nq", [(32, 9, 16), (8, 7, 9), (16, 8, 21), (32, 6, 5)])
def test_multi_query_pass_without_emd(ctx, dtype, k, nq):
    """When neither the model nor the requested statistics include the earth mover's distance, the Q x M pass runs its count-only
    form (prefix half of the mirror neither fetched nor scored); every other statistic and the model score are unchanged."""
    import json
    import os
    from golden_util import GOLDEN
    seqs, _ = synth.families(700 + k, 60, 1000, family=10)
    hs = api.HistogramSet(ctx, k, dtype, len(seqs))
    hs.build(seqs)
    fx = json.load(open(os.path.join(GOLDEN, "train_k7_u8_slow.json")))          # a reference-trained model without emd: euclidean, normalized_vectors, simratio
    text = "k: %d\nmode: 1\nmax_features: 3\nID: 0.8\nDatatype: uint%d_t\nfeature_set: 0\n" % (k, dtype) + fx["block"]
    feat = api.Feature.from_text(ctx, text, 0)
    mask = FAST_MASK & ~((1 << 7) | (1 << 29) | (1 << 18))
    cands = np.arange(2, len(seqs), dtype=np.uint32)
    qs = (np.arange(nq, dtype=np.uint32) * 2) % len(seqs)
    multi = api.score_multi(ctx, feat, hs, cands, hs, qs, feat_mask=mask)
    if nq >= (8 if dtype == 8 else 6 if dtype == 16 else 4) and k >= 6:
        assert "no emd" in ctx.last_kernel_info()[0]
    for i, q in enumerate(qs):
        single = feat.compute(hs, cands, hs, int(q))
        raw = api.pair_features_raw(ctx, hs, cands, hs, int(q), mask)
        assert np.array_equal(multi["raw"][i], raw), i
        assert np.array_equal(multi["sum"][i], single["sum"]) and np.array_equal(multi["csum"][i], single["csum"]), i


def test_cluster_driver_trains_its_own_model(tmp_path, ctx):
    """Without --recover the driver picks k by the reference's find_k rule and the narrowest histogram type, trains a model
    (own pair generator + msc_train_class), leaves it in weights.txt like the reference does, and clusters with it: every
    input sequence appears in exactly one cluster, and the weights file loads."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seqs, headers = synth.families(4242, 1200, 1000)
    fa = str(tmp_path / "in.fa")
    synth.write_fasta(fa, seqs, headers)
    out = str(tmp_path / "o.clstr")
    r = subprocess.run([os.path.join(root, "meshclust2_amd", "host", "msc_cluster"), fa, "--id", "0.9", "--output", out], cwd=str(tmp_path), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900)
    log = r.stdout.decode(errors="replace")
    assert r.returncode == 0, log[-2000:]
    # find_k measures blanks + sequence (test_find_k_reproduces_the_reference_rule): ceil(log4(~2000)) - 1 = 5, as the reference prints
    # for this kind of input; no count above 255 at k = 5 / 1 kb
    assert "Recommended K: 5" in log and "Using 8 bit histograms" in log
    members = [ln for ln in open(out) if not ln.startswith(">Cluster")]
    assert len(members) == len(seqs) and len({ln.split(">")[1].split("...")[0] for ln in members}) == len(seqs)
    feat = api.Feature.from_text(ctx, open(str(tmp_path / "weights.txt")).read(), 0)
    assert feat is not None


def test_cluster_driver_mixed_lengths_slow_features(tmp_path):
    """BASELINE cfg5 in small: mixed lengths, a `--feat slow` model that uses jensen_shannon, --id 0.6. The driver reproduces the
    reference CLI's .clstr byte for byte (the divergence statistics run through the per-centre path of the batched update)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seqs, hdrs = [], []
    for gi, (n, length, seed) in enumerate(((800, 600, 31), (800, 1000, 32), (800, 1500, 33))):
        s_, h_ = synth.families(seed, n, length, length_jitter=120)
        seqs += s_
        hdrs += [">m%d_%s" % (gi, x[1:]) for x in h_]
    seqs, hdrs = seqs[::3], hdrs[::3]
    fa = str(tmp_path / "mixed.fa")
    synth.write_fasta(fa, seqs, hdrs)
    out = str(tmp_path / "out.clstr")
    golden = os.path.join(root, "tests", "golden")
    r = subprocess.run([os.path.join(root, "meshclust2_amd", "host", "msc_cluster"), fa, "--recover", os.path.join(golden, "weights_mixed_slow_k6_u16.txt"),
                        "--id", "0.6", "--kmer", "6", "--datatype", "16", "--output", out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
    got, exp = open(out, "rb").read(), open(os.path.join(golden, "mixed_slow.clstr"), "rb").read()
    assert got == exp, "CLSTR differs: %d vs %d bytes" % (len(got), len(exp))


def _same_clusters(a, b):
    """two .clstr texts hold the same clusters (as sets of member headers), whatever the member / cluster order and the star"""
    def sets(text):
        out, cur = [], None
        for ln in text.decode().splitlines():
            if ln.startswith(">Cluster"):
                cur = set()
                out.append(cur)
            else:
                cur.add(ln.split(">")[1].split("...")[0])
        return sorted(sorted(c) for c in out)
    return sets(a) == sets(b)


@pytest.mark.parametrize("tag,run_cap,bits", [("cfg5", 900, 8), ("cfg5_u16", 3000, 16)])
@pytest.mark.parametrize("extra", [[], ["--sparse"], ["--serial-update"]])
def test_cfg5_mixed_lengths_slow_features_at_k9(tmp_path, tag, run_cap, bits, extra):
    """BASELINE.json configs[4] at its stated parameters, scaled to 240 sequences: k = 9, lengths log-uniform 500 .. 50 000
    (sequences on both sides of the 32 768-k-mer limit of the sort builder; tandem repeats push counts past the 8/16-entry
    divergence tables, in the second set past 255 so the reference picks 16-bit histograms), `--feat slow` (the model the
    reference trained uses jensen_shannon), `--id 0.6`. The fixture is the reference CLI's own .clstr; the driver, fed the model
    that run trained (and the histogram type recorded in it), reproduces it from the dense layout, the sparse layout and the
    centre-by-centre update order (predict/Feature.cpp:984-1009,1231-1263; clutil/Loader.cpp:42-86)."""
    import os
    import subprocess
    from golden_util import cfg5_set
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    golden = os.path.join(root, "tests", "golden")
    seqs, hdrs = cfg5_set(run_cap=run_cap)
    assert min(len(s_) for s_ in seqs) < 600 and max(len(s_) for s_ in seqs) > 40000
    fa = str(tmp_path / "cfg5.fa")
    synth.write_fasta(fa, seqs, hdrs)
    out = str(tmp_path / "out.clstr")
    r = subprocess.run([os.path.join(root, "meshclust2_amd", "host", "msc_cluster"), fa, "--recover", os.path.join(golden, "weights_%s_k9.txt" % tag), "--id", "0.6",
                        "--kmer", "9", "--output", out] + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200)
    log = r.stdout.decode(errors="replace")
    assert r.returncode == 0, log[-2000:]
    assert "Datatype: uint%d_t" % bits in weights_text("weights_%s_k9.txt" % tag)      # the type the reference chose for this set
    got, exp = open(out, "rb").read(), open(os.path.join(golden, "%s.clstr" % tag), "rb").read()
    assert _same_clusters(got, exp), "clusters differ"
    assert got == exp, "same clusters, but the CLSTR bytes differ: %d vs %d bytes" % (len(got), len(exp))


@pytest.mark.parametrize("extra", [[], ["--sparse"]])
def test_jitter_slow_windows_through_the_rank_form_divergences(tmp_path, extra):
    """900 sequences of 1 kb +- 100 (families of 10), k = 9, the `--feat slow` model the reference trained on them (it uses
    jefferey_divergence, predict/Feature.cpp:1231-1263), --id 0.8: every accumulate step scores a REAL length window through
    msc_get_close_window, whose divergence sums come from the rank form (k_pair_ranks_items counts cells, k_rank_items_finish adds them:
    the same terms as the merge kernels' in another order of addition). The fixture is the reference CLI's own .clstr;
    the driver reproduces it byte for byte from the dense layout (through the sparse mirror) and the sparse one -- and the library's own
    count says the passes did run over rank lists."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    golden = os.path.join(root, "tests", "golden")
    seqs, hdrs = synth.families(4711, 900, 1000, family=10, length_jitter=100)
    fa = str(tmp_path / "jitter.fa")
    synth.write_fasta(fa, seqs, hdrs)
    out = str(tmp_path / "out.clstr")
    r = subprocess.run([os.path.join(root, "meshclust2_amd", "host", "msc_cluster"), fa, "--recover", os.path.join(golden, "weights_jitter_slow_k9.txt"), "--id", "0.8",
                        "--kmer", "9", "--datatype", "8", "--output", out] + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200,
                       env=dict(os.environ, MSC_PROFILE_CALLS="1"))
    log = r.stdout.decode(errors="replace")
    assert r.returncode == 0, log[-2000:]
    got, exp = open(out, "rb").read(), open(os.path.join(golden, "jitter_slow.clstr"), "rb").read()
    assert _same_clusters(got, exp), "clusters differ"
    assert got == exp, "same clusters, but the CLSTR bytes differ: %d vs %d bytes" % (len(got), len(exp))
    m = re.search(r"list passes: (\d+) pairs scored inside their length windows \((\d+) of them over rank lists\)", log)
    assert m and int(m.group(2)) > 0.9 * int(m.group(1)) > 10000, log[-1500:]


@pytest.mark.parametrize("extra", [[], ["--sparse"]])
def test_cluster_driver_k9_uint8(tmp_path, extra):
    """BASELINE cfg3 in small: k = 9 with the histogram type the reference CLI chose by itself (uint8_t, 256 KiB histograms); the
    driver, fed the model that run trained, writes the same .clstr bytes from the dense and from the sparse layout."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seqs, hdrs = synth.families(61, 320, 1000, family=16)
    fa = str(tmp_path / "k9.fa")
    synth.write_fasta(fa, seqs, hdrs)
    out = str(tmp_path / "out.clstr")
    golden = os.path.join(root, "tests", "golden")
    r = subprocess.run([os.path.join(root, "meshclust2_amd", "host", "msc_cluster"), fa, "--recover", os.path.join(golden, "weights_k9_u8.txt"), "--id", "0.9",
                        "--output", out] + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
    got, exp = open(out, "rb").read(), open(os.path.join(golden, "k9_u8.clstr"), "rb").read()
    assert got == exp, "CLSTR differs: %d vs %d bytes" % (len(got), len(exp))


@pytest.mark.parametrize("ranks", [1, 2])
def test_cluster_driver_compacts_its_sparse_centre_store(tmp_path, monkeypatch, ranks):
    """The sparse centre store is append-only; when its arena runs out msc_cluster copies the live centres into a second store and
    clears the first for the next time (msc_hist_copy_batch + msc_hist_set_clear). With an arena that holds the centres little more
    than twice over, a small run compacts several times and still writes the golden .clstr bytes."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seqs, hdrs = synth.families(61, 320, 1000, family=16)
    fa = str(tmp_path / "k9.fa")
    synth.write_fasta(fa, seqs, hdrs)
    out = str(tmp_path / "out.clstr")
    golden = os.path.join(root, "tests", "golden")
    args = [fa, "--recover", os.path.join(golden, "weights_k9_u8.txt"), "--id", "0.9", "--output", out, "--sparse"]
    if ranks == 1:
        env = dict(os.environ, MSC_CLUSTER_CENTRE_ARENA="48000", MSC_CLUSTER_PROFILE="1")          # (r05: a set() to the point a centre already holds appends nothing: the arena is tighter than it was)
        r = subprocess.run([os.path.join(root, "meshclust2_amd", "host", "msc_cluster")] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, env=env)
        log = r.stdout.decode(errors="replace")
        assert r.returncode == 0, log[-2000:]
    else:          # every rank keeps the whole centre store (msc::GpuShardEngine): the same two standing stores there
        monkeypatch.setenv("MSC_CLUSTER_CENTRE_ARENA", "56000")       # (the engine moves a round's centres in one batch: room for the live lists + one round)
        monkeypatch.setenv("MSC_CLUSTER_PROFILE", "1")
        rcs, log, logs = _cluster_ranks(args, ranks, tmp_path, block=40)
        assert all(rc == 0 for rc in rcs), "\n".join(logs)[-3000:]
    m = re.search(r"centre store: rebuilt (\d+) times", log)
    assert m and int(m.group(1)) >= (2 if ranks == 1 else 1), log[-1500:]          # (r05: centres that stay on their point append nothing; the batch engine's arena cannot go lower)
    got, exp = open(out, "rb").read(), open(os.path.join(golden, "k9_u8.clstr"), "rb").read()
    assert got == exp, "CLSTR differs: %d vs %d bytes" % (len(got), len(exp))


def test_degenerate_inputs(ctx, oracle):
    """Empty candidate lists, a single candidate, no queries, sequences shorter than k, sequences of N only: the operators return
    what the reference's loops would (nothing scored, is_min true, best (-1, -1) / merge 0) instead of faulting."""
    k, dtype = 6, 32
    seqs, _ = synth.families(808, 12, 400, family=4)
    seqs = list(seqs) + [b"ACG", b"N" * 300, b"ACGTAC"]          # shorter than k, all N, exactly k
    hs = api.HistogramSet(ctx, k, dtype, len(seqs))
    hs.build(seqs)
    for i in (12, 13, 14):
        oh = oracle.hist(seqs[i], k, dtype)
        assert np.array_equal(hs.download(i), oh.array()) and hs.info(i)["length"] == oh.length
    feat = api.Feature.from_text(ctx, weights_text("weights_k9_u32.txt"), 0)
    trn = api.Trainer(ctx, feat, 0.9)
    none = np.zeros(0, dtype=np.uint32)
    flags, bp, bs, im = trn.get_close(hs, none, hs, 0)
    assert flags.size == 0 and bp == -1 and bs == -1.0 and im
    flags, bp, bs, im = trn.get_close(hs, np.array([3], dtype=np.uint32), hs, 3)
    of, obp, obs, oim = oracle.get_close(oracle.predictor(weights_text("weights_k9_u32.txt")), 0.9, oracle.hist(seqs[3], k, dtype), [oracle.hist(seqs[3], k, dtype)])
    assert np.array_equal(flags, of) and bp == obp and im == oim
    multi = api.score_multi(ctx, feat, hs, none, hs, np.arange(5, dtype=np.uint32))
    assert multi["sum"].shape == (5, 0)
    multi = api.score_multi(ctx, feat, hs, np.arange(12, dtype=np.uint32), hs, none)
    assert multi["sum"].shape == (0, 12)
    nearest, kept = trn.update_centres(hs, np.zeros(0, dtype=np.uint32), hs, [])
    assert nearest.size == 0
    nearest, kept = trn.update_centres(hs, np.array([0, 1], dtype=np.uint32), hs, [none, none])
    assert list(nearest) == [-1, -1] and list(kept) == [0, 0]
    assert list(trn.merge_all(hs, np.array([2], dtype=np.uint32), 5)) == [0]
    # a zero-length candidate never reaches length_difference in get_close: the length window drops it first (cluster/Trainer.cpp:39-48)
    flags, bp, bs, im = trn.get_close(hs, np.array([13], dtype=np.uint32), hs, 0)
    assert list(flags) == [0] and bp == -1 and im


@pytest.mark.parametrize("k,dtype,sparse", [(5, 16, False), (9, 32, False), (9, 32, True), (11, 8, True)])
def test_batches_without_a_single_base(ctx, oracle, k, dtype, sparse):
    """A batch in which no sequence contributes a base -- all records empty, or (fastcar's string overload, strip = 1,
    clutil/Loader.cpp:115-121) a soft-masked lower-case chunk of which the strip leaves nothing: the packed stream is then
    0 bytes long. The slots must come out as the reference's points do (every bin the pseudocount), dense and sparse; a later
    ordinary batch into the same set still builds correctly."""
    hs = api.HistogramSet(ctx, k, dtype, 6, sparse_entries=5000 if sparse else 0)
    hs.build([b"", b"", b""])
    lower = [b"acgtacgtacgtnnnnacgtacgatcgatcgatcagctacgatcgactagc" * 4, b"nnnnnnnn", b"acgt"]
    hs.build(lower, first_slot=3, strip=True)
    for i, s_ in enumerate([b"", b"", b""] + lower):
        oh = oracle.hist(s_, k, dtype, strip=i >= 3)
        inf = hs.info(i)
        assert (inf["mag"], inf["length"], inf["one_mers"], inf["overflow"]) == (oh.mag, oh.length, list(oh.one_mers), oh.overflow), i
        if k <= 9:
            assert np.array_equal(hs.download(i), oh.array()), i
        if sparse:
            assert hs.entries(i) == 0
    seqs, _ = synth.families(5150 + k, 3, 700, family=3)
    hs.build(seqs, first_slot=1)
    for i, s_ in enumerate(seqs):
        oh = oracle.hist(s_, k, dtype)
        assert hs.info(1 + i)["mag"] == oh.mag
        if k <= 9:
            assert np.array_equal(hs.download(1 + i), oh.array())


@pytest.mark.parametrize("dtype,k,wts", [(16, 5, "weights_k5_u16.txt"), (32, 9, "weights_k9_u32.txt")])
def test_centre_gather_in_place_keeps_the_merge_scan(ctx, dtype, k, wts):
    """SURVEY 8(e), update round: shard.ShardedCentres moves centre slots through torch views of the set's device memory
    (msc_hist_set_device_view -> payload rows -> gathered rows -> msc_hist_import_done) exactly as the RCCL all-gather does
    on N ranks; Trainer::merge over the gathered slots == over the original ones, also after a Q x M pass has built the
    digest mirror (the gathered slots must be re-digested)."""
    import torch
    from meshclust2_amd import shard
    seqs, _ = synth.families(515 + k, 60, 700, family=6)
    pts = api.HistogramSet(ctx, k, dtype, len(seqs))
    pts.build(seqs)
    feat = api.Feature.from_text(ctx, weights_text(wts), 0)
    trn = api.Trainer(ctx, feat, 0.9)
    nc = 23
    plan = shard.ShardPlan(nc, 1, block=4)
    n_pad = plan.local_count(0)
    cen = api.HistogramSet(ctx, k, dtype, 2 * n_pad)          # slots [0, n_pad): own centres; [n_pad, 2 n_pad): gathered
    for c in range(nc):
        cen.clone_from(c, pts, c)
    own = np.arange(nc, dtype=np.uint32)
    # a Q x M pass over the (still empty) gathered half builds the digest mirror with stale contents for those slots
    for c in range(nc):
        cen.clone_from(n_pad + c, pts, len(seqs) - 1 - c)
    before = api.score_multi(ctx, feat, cen, np.arange(n_pad, n_pad + nc, dtype=np.uint32), cen, np.arange(16, dtype=np.uint32), want=("sum",))["sum"]
    ctx.synchronize()
    bins, scal = shard.device_tensors(cen, 2 * n_pad)

    class Backend:
        def centre_payload(self, n):
            return [bins[:n], scal[:n]]

        def gather_buffers(self, n_rows):
            return [bins[n_pad:n_pad + n_rows], scal[n_pad:n_pad + n_rows]]

        def import_centres(self, rows):
            torch.cuda.synchronize()
            cen.import_done(n_pad, n_pad)

    rows = shard.ShardedCentres(None, plan, Backend(), 0, device="cuda").gather()
    assert rows.tolist() == list(range(nc))
    gathered = (n_pad + rows).astype(np.uint32)
    for delta in (1, 5):
        assert list(trn.merge_all(cen, gathered, delta)) == list(trn.merge_all(cen, own, delta))
    a = api.score_multi(ctx, feat, cen, gathered, cen, np.arange(16, dtype=np.uint32), want=("sum",))["sum"]
    b = api.score_multi(ctx, feat, cen, own, cen, np.arange(16, dtype=np.uint32), want=("sum",))["sum"]
    assert np.array_equal(a, b) and not np.array_equal(a, before)


@pytest.mark.parametrize("dtype,k", [(8, 8), (16, 8), (32, 9), (64, 8), (8, 10)])
def test_sort_build_length_classes_saturation_prefixes(ctx, oracle, dtype, k):
    """The large-k builder (k_build_sort: LDS sort + one streaming write per slot): sequences of all four LDS classes in one
    batch, empty and N-only records, runs that saturate uint8_t (>= 255 equal k-mers, several per sequence, in different
    tiles), and the tile prefixes / sums it derives from the sorted runs (checked through emd and the other statistics).
    A batch holding a sequence with more than 32768 k-mers takes the fill + count + finalize path: same answers."""
    rng = np.random.default_rng(31 * k + dtype)
    rnd = lambda n: bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n))
    seqs = [rnd(900), rnd(3000), rnd(12000), rnd(30000), b"", b"N" * 50, b"ACG", rnd(200) + b"N" * 40 + rnd(300),
            b"A" * 700 + rnd(400) + b"C" * 600 + rnd(100) + b"GT" * 500,          # saturating runs for u8 in three tiles
            b"AC" * 10000 + rnd(50), rnd(1020 + k), rnd(4000 + k)]
    long_batch = seqs[:4] + [b"AC" * 20000 + rnd(100)]
    for batch in (seqs, long_batch):
        hs = api.HistogramSet(ctx, k, dtype, len(batch))
        hs.build(batch)
        oh = [oracle.hist(s, k, dtype) for s in batch]
        for i in range(len(batch)):
            assert np.array_equal(hs.download(i), oh[i].array()), i
            inf = hs.info(i)
            assert (inf["mag"], inf["length"], inf["one_mers"], inf["overflow"]) == (oh[i].mag, oh[i].length, list(oh[i].one_mers), oh[i].overflow), i
        scored = [i for i in range(len(batch)) if oh[i].length > 0]
        for q in scored[:3] + scored[-2:]:
            raw = api.pair_features_raw(ctx, hs, np.array(scored, dtype=np.uint32), hs, q, FAST_MASK, api.ORDER_CAND_FIRST)
            for i, c in enumerate(scored):
                for col, (name, bit) in zip(raw[i], FAST):
                    exp = oracle.raw_feature(1 << bit, oh[c], oh[q])
                    if name in EXACT and name != "kulczynski2":
                        assert col == exp or (np.isnan(col) and np.isnan(exp)), (name, c, q)
                    else:
                        assert col == pytest.approx(exp, rel=RTOL, abs=1e-13, nan_ok=True), (name, c, q)      # an all-ones histogram has no variance


@pytest.mark.parametrize("strip", [False, True])
def test_threaded_host_encode_matches_oracle(ctx, oracle, strip):
    """msc_hist_build encodes on several host threads once a batch holds >= 512 K characters (each sequence then starts on a
    byte boundary of the packed stream): ragged lengths, N runs that merge or split segments, lower case, empty records and
    the strip overload must come out exactly as the oracle's one-by-one encoding."""
    rng = np.random.default_rng(5 + int(strip))
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = []
    for i in range(900):
        n = int(rng.integers(0, 1800))
        s = bytearray(rng.choice(alpha, n).tobytes())
        if n > 100 and i % 3 == 0:
            for _ in range(int(rng.integers(1, 4))):
                a = int(rng.integers(0, n - 50))
                g = int(rng.integers(1, 30))
                s[a:a + g] = b"N" * g
        if i % 7 == 0:
            s = bytearray(bytes(s).lower())
        seqs.append(bytes(s))
    assert sum(len(s) for s in seqs) > (1 << 19)
    hs = api.HistogramSet(ctx, 5, 16, len(seqs))
    hs.build(seqs, strip=strip)
    for i, s in enumerate(seqs):
        o = oracle.hist(s, 5, 16, strip)
        assert np.array_equal(hs.download(i), o.array()), i
        inf = hs.info(i)
        assert (inf["length"], inf["one_mers"], inf["mag"]) == (o.length, list(o.one_mers), o.mag), i


@pytest.mark.parametrize("seed", [3, 11, 42, 77, 1115, 1149])
def test_randomised_rounds(ctx, seed):
    """A few fixed seeds of tests/fuzz_parity.py (random k, datatype, layout, alphabets, lengths; builders, the 11 statistics in both
    argument orders, get_close / filter / merge / mean_nearest under a fixture model, error parity where the reference throws).
    Seed 1115 holds a pair whose normalisation is NaN: the library must fail exactly where the reference does."""
    import fuzz_parity
    assert "ok" in fuzz_parity.run_round(ctx, seed)


@pytest.mark.parametrize("seed", [80000, 80004, 80008, 80010, 80019, 80025, 80027, 80046])      # digest (u8/u16 counts, with/without emd), ring, register, sparse-queued
def test_randomised_multi_rounds(ctx, seed):
    """Fixed seeds of tests/fuzz_parity.py's Q x M rounds: msc_score_multi (whichever kernel the library picks for the random
    k / datatype / layout / query count / model) against one 1 x M pass per query."""
    import fuzz_parity
    assert "ok" in fuzz_parity.run_multi_round(ctx, seed)


def test_fastcar_sparse_layout_writes_the_same_file(tmp_path):
    """msc_fastcar --sparse (sorted (bin, value) lists; the Q x M pass queues one merge-path pass per query) writes the same bytes
    as the dense layout, also when the database and the queries come in several chunks (a sparse set is rebuilt per chunk)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seqs, headers = synth.families(123, 260, 1000, length_jitter=120)
    db, q = str(tmp_path / "db.fa"), str(tmp_path / "q.fa")
    synth.write_fasta(db, seqs, headers)
    synth.write_fasta(q, seqs[3:120:4], headers[3:120:4])
    outs = []
    for extra in ([], ["--sparse"], ["--sparse", "--chunk", "100"]):
        prefix = str(tmp_path / ("fc%d_" % len(outs)))
        r = subprocess.run([os.path.join(root, "meshclust2_amd", "host", "msc_fastcar"), db, "--query", q, "--recover", os.path.join(root, "tests", "golden", "weights_k9_u32.txt"),
                            "--output", prefix] + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
        outs.append(open(prefix + "0", "rb").read())
    assert outs[0] == outs[1] and outs[0].count(b"\n") > 100
    # chunking changes the order of the lines (query chunk, then database chunk), not their set
    assert sorted(outs[2].splitlines()) == sorted(outs[0].splitlines())


def test_find_k_reproduces_the_reference_rule(tmp_path):
    """Runner::find_k (cluster/CRunner.cpp:479-502) measures every record through ChromListMaker::makeChromList, whose pre-sized
    Chromosome is APPENDED to: the record size it averages is that of `len` blanks + the sequence (so 2 x len without N runs).
    Expected (avg length, k) pairs were printed by the reference CLI for inputs of exactly these shapes (lengths and N runs are all
    that matter): 100 -> 200/3; 100,300 -> 400/4; 5 x 1000 -> 2000/5; 5 x 1000 + 50 -> 1683/5; 100 N30 100 -> 430/4;
    100 N5 100 -> 410/4; 100 N30 10 -> 240/3; N20 100 N20 -> 240/3; 15 -> 30/2; two files (100 | 100, 300) -> 300/4."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "meshclust2_amd", "host", "msc_cluster")
    b = lambda n: ("ACGTTGCA" * (n // 8 + 1))[:n]
    cases = [([[b(100)]], 200, 3), ([[b(100), b(300)]], 400, 4), ([[b(1000)] * 5], 2000, 5), ([[b(1000)] * 5 + [b(50)]], 1683, 5),
             ([[b(100) + "N" * 30 + b(100)]], 430, 4), ([[b(100) + "n" * 5 + b(100)]], 410, 4), ([[b(100) + "N" * 30 + b(10)]], 240, 3),
             ([["N" * 20 + b(100) + "N" * 20]], 240, 3), ([[b(15)]], 30, 2), ([[b(100)], [b(100), b(300)]], 300, 4)]
    for ci, (files, avg, k) in enumerate(cases):
        names = []
        for fi, recs in enumerate(files):
            name = str(tmp_path / ("c%d_%d.fa" % (ci, fi)))
            with open(name, "w") as f:
                for ri, s in enumerate(recs):
                    f.write(">r%d_%d\n" % (fi, ri))
                    for a in range(0, len(s), 70):
                        f.write(s[a:a + 70] + "\n")
            names.append(name)
        r = subprocess.run([exe] + names + ["--id", "0.9", "--output", str(tmp_path / "o.clstr")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        out = r.stdout.decode(errors="replace")          # too few sequences to train on: the run stops after printing its choice of k
        assert "avg length: %d\n" % avg in out and "Recommended K: %d\n" % k in out, (ci, out[-400:])


def test_operators_without_kernel_timing(oracle):
    """msc_set_kernel_timing(0) (what the step-serial drivers do: four event records per 1 x M call cost 18 us): every operator
    gives the same answers, msc_last_kernel_ms reports that nothing was timed, and switching back on works."""
    ctx = api.Context(0)
    seqs, _ = synth.families(4242, 40, 900, family=8)
    hs = api.HistogramSet(ctx, 8, 16, len(seqs))
    hs.build(seqs)
    feat = api.Feature.from_text(ctx, weights_text("weights_k8_u16.txt"), 0)
    trn = api.Trainer(ctx, feat, 0.9)
    w = np.arange(1, len(seqs), dtype=np.uint32)
    want = (trn.get_close(hs, w, hs, 0), trn.filter(hs, 3, hs, w), trn.closest(hs, w[:9])[0], api.score_multi(ctx, feat, hs, w, hs, [0, 5, 9, 11])["sum"])
    assert ctx.last_kernel_ms()[0] > 0
    ctx.set_kernel_timing(False)
    got = (trn.get_close(hs, w, hs, 0), trn.filter(hs, 3, hs, w), trn.closest(hs, w[:9])[0], api.score_multi(ctx, feat, hs, w, hs, [0, 5, 9, 11])["sum"])
    with pytest.raises(api.MscError):
        ctx.last_kernel_ms()
    assert np.array_equal(want[0][0], got[0][0]) and want[0][1:] == got[0][1:]
    assert np.array_equal(want[1], got[1]) and want[2] == got[2] and np.array_equal(want[3], got[3])
    assert trn.merge(hs, w, 2, 3, 8) == trn.merge(hs, w, 2, 3, 8)
    ctx.set_kernel_timing(True)
    trn.get_close(hs, w, hs, 0)
    assert ctx.last_kernel_ms()[0] > 0
    ctx.close()


def _run_ranks(module_args, n, tmp_path, timeout=900):
    """n ranks of `python -m torch.distributed.run ...` sharing GPU 0 over gloo (MSC_BENCH_ONE_GPU: this pool has 1-GPU boxes; the
    real launch is one rank per GPU over RCCL)"""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    env = dict(os.environ, MSC_BENCH_BACKEND="gloo", MSC_BENCH_ONE_GPU="1", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                           "--master-port", str(port)] + module_args, cwd=str(tmp_path), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)


def _cluster_ranks(args, ranks, tmp_path, block=100, timeout=900):
    """`ranks` processes of msc_cluster, each holding 1 / ranks of the points (msc::ShardedBackend over msc::GpuShardEngine), sharing
    this box's one GPU (MSC_ONE_GPU) and exchanging over plain sockets with host staging (MSC_COMM=tcp) -- the pool has 1-GPU boxes;
    the production launch is one rank per GPU over RCCL. -> (return codes, rank 0's log)"""
    import os
    import socket
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    procs = []
    for r in range(ranks):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(ranks), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MSC_COMM="tcp", MSC_ONE_GPU="1",
                   MSC_SHARD_BLOCK=str(block))
        procs.append(subprocess.Popen([os.path.join(root, "meshclust2_amd", "host", "msc_cluster")] + args, cwd=str(tmp_path), env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    logs = []
    for p_ in procs:
        try:
            logs.append(p_.communicate(timeout=timeout)[0].decode(errors="replace"))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    return [p_.returncode for p_ in procs], logs[0], logs


def _step_collectives(log):
    import re
    m = re.search(r"get_close steps (\d+) collectives (\d+) overflow (\d+)", log)
    assert m, log[-1500:]
    return tuple(int(x) for x in m.groups())


@pytest.mark.parametrize("case,ranks,extra", [("cfg1", 2, []), ("cfg1", 3, []), ("k9_u8", 2, []), ("k9_u8", 2, ["--sparse"]), ("k9_u8", 3, ["--sparse"]),
                                              ("cfg5", 2, ["--sparse"]), ("cfg5", 3, ["--sparse"]), ("cfg5", 2, []), ("cfg5_u16", 3, ["--sparse"])])
def test_multi_rank_cluster_driver_on_the_gpu(tmp_path, case, ranks, extra):
    """msc_cluster with the points sharded over `ranks` processes: every rank runs the clustering logic of msc_driver.hpp on replicated
    bookkeeping and scores its shard through the C ABI, dense or sparse; the query travels as a packed slot, get_mean as column sums.
    Rank 0 writes the reference CLI's own .clstr byte for byte (cluster/ClusterFactory.cpp:553-656) -- cfg1, the k = 9 / uint8_t set
    (BASELINE cfg3's shape), and BASELINE cfg5's parameters (k = 9, 500 b - 50 kb, a `--feat slow` model, --id 0.6) -- and a get_close
    step costs at most one broadcast and one all-gather."""
    import os
    from golden_util import cfg5_set
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    golden = os.path.join(root, "tests", "golden")
    if case.startswith("cfg5"):
        seqs, hdrs = cfg5_set(run_cap=900 if case == "cfg5" else 3000)
        args = ["--recover", os.path.join(golden, "weights_%s_k9.txt" % case), "--id", "0.6", "--kmer", "9"]
        clstr, block = case + ".clstr", 16
    else:
        seed, n, fam, wts, clstr = {"cfg1": (20260001, 1000, 20, "weights_k5_u16.txt", "cfg1.clstr"), "k9_u8": (61, 320, 16, "weights_k9_u8.txt", "k9_u8.clstr")}[case]
        seqs, hdrs = synth.families(seed, n, 1000, family=fam)
        args = ["--recover", os.path.join(golden, wts), "--id", "0.9"] + (["--kmer", "5", "--datatype", "16"] if case == "cfg1" else [])
        block = 100 if case == "cfg1" else 40
    fa = str(tmp_path / "in.fa")
    synth.write_fasta(fa, seqs, hdrs)
    out = str(tmp_path / "out.clstr")
    rcs, log, logs = _cluster_ranks([fa] + args + ["--output", out] + extra, ranks, tmp_path, block=block)
    assert all(rc == 0 for rc in rcs), "\n".join(logs)[-3000:]
    assert open(out, "rb").read() == open(os.path.join(golden, clstr), "rb").read()
    steps, coll, overflow = _step_collectives(log)
    assert steps > 0 and coll <= 2 * steps + overflow


def _cluster_case(case, tmp_path):
    """(arguments, fixture name, shard block) of a .clstr fixture case for msc_cluster; writes in.fa"""
    import os
    from golden_util import cfg5_set
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    golden = os.path.join(root, "tests", "golden")
    if case.startswith("cfg5"):
        seqs, hdrs = cfg5_set(run_cap=900 if case == "cfg5" else 3000)
        args = ["--recover", os.path.join(golden, "weights_%s_k9.txt" % case), "--id", "0.6", "--kmer", "9"]
        clstr, block = case + ".clstr", 16
    else:
        seed, n, fam, wts, clstr = {"cfg1": (20260001, 1000, 20, "weights_k5_u16.txt", "cfg1.clstr"), "k9_u8": (61, 320, 16, "weights_k9_u8.txt", "k9_u8.clstr")}[case]
        seqs, hdrs = synth.families(seed, n, 1000, family=fam)
        args = ["--recover", os.path.join(golden, wts), "--id", "0.9"] + (["--kmer", "5", "--datatype", "16"] if case == "cfg1" else [])
        block = 100 if case == "cfg1" else 40
    fa = str(tmp_path / "in.fa")
    synth.write_fasta(fa, seqs, hdrs)
    return [fa] + args, os.path.join(golden, clstr), block


@pytest.mark.parametrize("case,extra", [("k9_u8", []), ("k9_u8", ["--sparse"]), ("cfg5", ["--sparse"]), ("cfg5", []), ("cfg1", [])])
def test_rank_driver_over_rccl_on_one_gpu(tmp_path, case, extra):
    """The RCCL transport of the rank driver (host/msc_comm_rccl.hpp) EXECUTED: msc_cluster with MSC_FORCE_SHARDED and WORLD_SIZE = 1
    takes ShardedBackend + RcclComm with the one-rank shortcuts off, so ncclCommInitRankConfig (non-blocking), the in-place
    ncclAllGather, ncclBroadcast, ncclAllReduce(uint64) and the host staging all run on the ctx stream of a real GPU -- and the run
    writes the reference CLI's .clstr byte for byte. (Every other sharded test moves its bytes over sockets: MSC_COMM=tcp.)"""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args, fixture, block = _cluster_case(case, tmp_path)
    out = str(tmp_path / "out.clstr")
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29711", MSC_FORCE_SHARDED="1", MSC_SHARD_BLOCK=str(block),
               MSC_COMM_TIMEOUT_S="120")
    env.pop("MSC_COMM", None)
    r = subprocess.run([os.path.join(root, "meshclust2_amd", "host", "msc_cluster")] + args + ["--output", out] + extra, cwd=str(tmp_path), env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    log = r.stdout.decode(errors="replace")
    assert r.returncode == 0, log[-3000:]
    assert open(out, "rb").read() == open(fixture, "rb").read()
    m = re.search(r"collectives: broadcast (\d+) all_gather (\d+) all_reduce (\d+)", log)
    assert m, log[-1500:]
    bc, ag, ar = (int(x) for x in m.groups())
    assert ag > 0 and (bc > 0 or ar > 0), log[-1500:]          # the collectives ran through RCCL, they were not short-cut
    steps, coll, overflow = _step_collectives(log)
    assert steps > 0 and coll <= 2 * steps + overflow


def test_rank_driver_over_rccl_on_two_gpus(tmp_path):
    """Two ranks of msc_cluster on two GPUs over RCCL (xGMI) -- only where the box has two (this pool's boxes have one: skipped there)."""
    import os
    import socket
    import subprocess
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args, fixture, block = _cluster_case("k9_u8", tmp_path)
    out = str(tmp_path / "out.clstr")
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MSC_SHARD_BLOCK=str(block),
                   MSC_COMM_TIMEOUT_S="120")
        env.pop("MSC_COMM", None)
        env.pop("MSC_ONE_GPU", None)
        procs.append(subprocess.Popen([os.path.join(root, "meshclust2_amd", "host", "msc_cluster")] + args + ["--output", out, "--sparse"], cwd=str(tmp_path), env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p_ in procs:
        try:
            logs.append(p_.communicate(timeout=600)[0].decode(errors="replace"))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p_.returncode == 0 for p_ in procs), "\n".join(logs)[-3000:]
    assert open(out, "rb").read() == open(fixture, "rb").read()


@pytest.mark.parametrize("ranks,extra", [(2, []), (3, []), (2, ["--sparse"]), (3, ["--sparse"])])
def test_multi_rank_cluster_driver_mixed_lengths(tmp_path, ranks, extra):
    """The fixtures above hold few real windows; here the lengths spread over 900-1100 bases, so every step scores a real window,
    clusters are opened and moved thousands of times, and the exchanges of every operator carry data (k = 9, 8-bit bins: BASELINE
    cfg3's shape). The one-GPU run of the same binary (itself held to the reference CLI by the fixtures and the fuzz) is the
    yardstick: same bytes, dense and sparse, 2 and 3 ranks with uneven shares."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    golden = os.path.join(root, "tests", "golden")
    seqs, hdrs = synth.families(777, 3000, 1000, length_jitter=100)
    fa = str(tmp_path / "in.fa")
    synth.write_fasta(fa, seqs, hdrs)
    common = [fa, "--recover", os.path.join(golden, "weights_k9_u8.txt"), "--id", "0.9", "--kmer", "9", "--datatype", "8"] + extra + ["--output"]
    one = str(tmp_path / "one.clstr")
    subprocess.check_call([os.path.join(root, "meshclust2_amd", "host", "msc_cluster")] + common + [one], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = str(tmp_path / "ranks.clstr")
    rcs, log, logs = _cluster_ranks(common + [out], ranks, tmp_path, block=128)
    assert all(rc == 0 for rc in rcs), "\n".join(logs)[-3000:]
    a, b = open(one, "rb").read(), open(out, "rb").read()
    assert a.count(b">Cluster") > 100 and a == b
    steps, coll, overflow = _step_collectives(log)
    assert steps > 500 and coll <= 2 * steps + overflow


@pytest.mark.parametrize("ranks", [2, 3])
def test_multi_rank_cluster_driver_k13_u64_sparse(tmp_path, ranks):
    """BASELINE cfg4's parameters over several ranks: k = 13, 64-bit counts, 20 kb sequences, the sparse layout (no dense form exists:
    512 MiB per histogram) -- the query travels as a ~240 KB list, get_mean as each rank's summed excess list. 2 and 3 ranks write the
    bytes of the one-GPU run."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    golden = os.path.join(root, "tests", "golden")
    seqs, hdrs = synth.families(4421, 120, 20000, family=6, length_jitter=2500)
    fa = str(tmp_path / "in.fa")
    synth.write_fasta(fa, seqs, hdrs)
    common = [fa, "--recover", os.path.join(golden, "weights_cfg4_k13.txt"), "--id", "0.9", "--kmer", "13", "--datatype", "64", "--sparse", "--output"]
    one = str(tmp_path / "one.clstr")
    subprocess.check_call([os.path.join(root, "meshclust2_amd", "host", "msc_cluster")] + common + [one], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = str(tmp_path / "ranks.clstr")
    rcs, log, logs = _cluster_ranks(common + [out], ranks, tmp_path, block=8)
    assert all(rc == 0 for rc in rcs), "\n".join(logs)[-3000:]
    a, b = open(one, "rb").read(), open(out, "rb").read()
    assert a == b and a.count(b">Cluster") >= 2, a.count(b">Cluster")


@pytest.mark.parametrize("exchange", ["sequences", "histograms"])
def test_bench_two_ranks_packed_exchange(tmp_path, exchange):
    """bench.py --gpus 2 (strong scaling: the sequences split over the ranks, 2 all-gathers per step assemble the query block -- as
    2-bit packed sequences every rank builds the histograms of, or as the histograms themselves) on two ranks sharing this GPU,
    against ONE rank scoring the same blocks (--check-world 2): with --check both lines carry, per timed step, the number of close
    candidates of every query of the block summed over the ranks -- the sharded run scores the same pairs and reaches the same
    decisions as one rank does."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--nseq", "8000", "--steps", "3", "--warmup", "1", "--queries", "16", "--cpu-seconds", "0", "--check", "--shard", "candidates", "--exchange", exchange]
    r = _run_ranks([os.path.join(root, "bench.py"), "--gpus", "2"] + common, 2, tmp_path)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-3000:]
    line = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["pairs_per_step"] == 16 * 8000
    assert line["roofline"]["candidates_per_launch"] == 4000 and line["value"] > 0
    assert line["config"]["workload"].startswith("custom") and "1kb" in line["metric"]          # 8000 sequences are not cfg2
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--check-world", "2"] + common, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert one.returncode == 0, one.stdout.decode(errors="replace")[-3000:]
    ref_line = json.loads([ln for ln in one.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert len(line["check"]) == 3 and all(len(c) == 16 for c in line["check"])
    assert line["check"] == ref_line["check"]
    assert sum(sum(c) for c in line["check"]) >= 3 * 16          # every query is at least close to itself


def test_bench_two_ranks_row_shards(tmp_path):
    """bench.py --gpus 2 as it runs by default since r05 (--shard rows): the candidates replicated by ONE set-up exchange (every rank
    generates half of the sequences, all-gathers the 2-bit rows and builds all 6 000 histograms from the gathered device buffer),
    each rank scoring half of every step's query rows against ALL candidates -- fastcar's own cut of the job
    (fastcar/FC_Runner.cpp:585-597). Two ranks sharing this GPU against ONE rank running the very same command: with --check both
    lines carry the per-query close counts of every timed step, which must be equal -- the same pairs, the same decisions."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--nseq", "6000", "--steps", "3", "--warmup", "1", "--queries", "272", "--cpu-seconds", "0", "--check", "--no-secondary"]
    r = _run_ranks([os.path.join(root, "bench.py"), "--gpus", "2"] + common, 2, tmp_path)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-3000:]
    line = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["pairs_per_step"] == 272 * 6000
    assert line["config"]["sharding"].startswith("query rows") and line["roofline"]["kernel"].startswith("k_pair_gemm_fp4_dma")
    assert line["roofline"]["candidates_per_launch"] == 6000          # every rank scores its rows against ALL candidates
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert one.returncode == 0, one.stdout.decode(errors="replace")[-3000:]
    ref_line = json.loads([ln for ln in one.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert len(line["check"]) == 3 and all(len(c) == 272 for c in line["check"])
    assert line["check"] == ref_line["check"]
    assert sum(sum(c) for c in line["check"]) >= 3 * 272          # every query is at least close to itself


@pytest.mark.parametrize("dtype,k,wts,sparse,n", [(16, 5, "weights_k5_u16.txt", False, 3000), (32, 9, "weights_k9_u32.txt", False, 400), (8, 9, "weights_k9_u8.txt", True, 2500),
                                                  (16, 9, "weights_cfg5_k9.txt", True, 600), (16, 5, "weights_k5_u16.txt", False, 140000)])
def test_get_close_over_a_device_window_equals_get_close_over_a_slot_list(ctx, dtype, k, wts, sparse, n):
    """msc_get_close_window (the accumulate loop's window kept on the device: cluster/ClusterFactory.cpp:553-610 over
    cluster/Trainer.cpp:23-71) against msc_get_close on the slot list of the same alive positions: the same close set, arg-max and
    is_min on random ranges while positions die (close ones by themselves, others through msc_window_kill), single-block and
    multi-block compaction, a `--feat slow` model, dense and sparse sets."""
    rng = np.random.default_rng(5 + k)
    seqs, _ = synth.families(900 + k, n, 1000, family=25, length_jitter=120)
    hs = api.HistogramSet(ctx, k, dtype, n, sparse_entries=(sum(len(s) for s in seqs) + 4096) if sparse else 0)
    for off in range(0, n, 1024):
        hs.build(seqs[off:off + 1024], first_slot=off)
    feat = api.Feature.from_text(ctx, weights_text(wts), 0)
    trn = api.Trainer(ctx, feat, 0.9 if "cfg5" not in wts else 0.6)
    order = rng.permutation(n).astype(np.uint32)
    win = api.Window(ctx, hs, order)
    alive = np.ones(n, dtype=bool)
    assert win.alive() == n
    for step in range(40 if n < 100000 else 8):
        q = int(rng.integers(n))
        a, b = sorted(int(x) for x in rng.integers(0, n + 1, 2))
        if step % 7 == 0:
            a, b = 0, n          # the whole store: more than 128 * 1024 positions take the multi-block compaction (the 140 000-point case)
        pos = a + np.flatnonzero(alive[a:b])
        assert win.alive(a, b) == pos.size
        flags, bp, bs, im = trn.get_close(hs, order[pos], hs, q) if pos.size else (np.zeros(0, dtype=np.uint8), -1, -1.0, True)
        close, wbp, wbs, wim = win.get_close(trn, a, b, hs, q)
        assert np.array_equal(close, pos[np.flatnonzero(flags)]), step
        assert wbp == (int(pos[bp]) if bp >= 0 else -1) and wim == im and (bp < 0 or wbs == bs), step
        alive[close] = False
        extra = np.flatnonzero(alive)
        if extra.size:
            kill = rng.choice(extra, size=min(extra.size, int(rng.integers(1, 9))), replace=False)
            win.kill(kill)
            alive[kill] = False
    with pytest.raises(api.MscError):
        win.kill(np.flatnonzero(~alive)[:1])          # a position dies once
