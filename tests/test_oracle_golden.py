"""The travelling CPU oracle (oracle/msc_oracle.c) against the committed golden fixtures, which were produced by the
real reference (tests/golden/gen_golden.py). Runs anywhere (no GPU, no /root/reference)."""
import numpy as np
import pytest

from golden_util import EXACT, FEATS, VECTOR_SETS, cfg4_sequences, dense_bins, kat, load_vectors, long_fragment_set, weights_text


def test_appendix_d_known_answers(oracle):
    k = kat()
    for dt in (8, 16, 32, 64):
        g = k["u%d" % dt]
        a, b = oracle.hist(k["A"], 2, dt), oracle.hist(k["B"], 2, dt)
        assert a.array().tolist() == g["hist_A"] and b.array().tolist() == g["hist_B"]
        assert list(a.one_mers) == g["one_mers_A"]
        assert oracle.lib().orc_distance(a, b) == g["distance"]
        for name, bit in FEATS:
            val = oracle.raw_feature(1 << bit, a, b)
            if name in EXACT:
                assert val == g["raw"][name], (dt, name)
            else:
                assert val == pytest.approx(g["raw"][name], rel=1e-12), (dt, name)
    # the hand-checkable numbers of SURVEY.md Appendix D
    assert k["u16"]["hist_A"] == [1, 5, 2, 4, 2, 2, 6, 1, 5, 1, 1, 4, 3, 3, 2, 3]
    assert k["u16"]["raw"]["manhattan"] == 7 and k["u16"]["raw"]["emd"] == 26
    assert k["u32"]["raw"]["simratio"] < 1e-6 < 0.98 < k["u16"]["raw"]["simratio"]


@pytest.mark.parametrize("vec,wts", VECTOR_SETS)
def test_vectors(oracle, vec, wts):
    v = load_vectors(vec)
    k, dt, n = int(v["k"]), int(v["dtype"]), int(v["n"])
    hs = [oracle.hist(bytes(s), k, dt) for s in v["seqs"]]
    for i, h in enumerate(hs):
        assert np.array_equal(h.array(), dense_bins(v, i))
        assert (h.mag, h.length) == (int(v["mag"][i]), int(v["length"][i]))
        assert list(h.one_mers) == v["one_mers"][i].tolist()
        assert h.stddev == pytest.approx(float(v["stddev"][i]), rel=1e-13)
    for i in range(n):
        for j in range(n):
            for f, (name, bit) in enumerate(FEATS):
                val = oracle.raw_feature(1 << bit, hs[i], hs[j])
                if name in EXACT:
                    assert val == v["raw"][i, j, f], (name, i, j)
                else:
                    assert val == pytest.approx(v["raw"][i, j, f], rel=1e-11, abs=1e-15), (name, i, j)
    pred = oracle.predictor(weights_text(wts))
    for i in range(n):
        for j in range(n):
            s, _, w = oracle.score(pred.cls, hs[i], hs[j])
            assert np.allclose(s, v["singles"][i, j], rtol=1e-11, atol=1e-14)
            assert w == pytest.approx(v["sums"][i, j], rel=1e-10, abs=1e-12)
            assert oracle.lib().orc_classify(pred, hs[i], hs[j]) == pytest.approx(v["csums"][i, j], rel=1e-11)
            assert oracle.lib().orc_p_predict(pred, hs[i], hs[j]) == pytest.approx(v["predict"][i, j], rel=1e-10, abs=1e-12)
            assert oracle.lib().orc_p_close(pred, hs[i], hs[j]) == int(v["close"][i, j])
    for ci, cutoff in enumerate(v["cutoffs"]):
        for q in range(n):
            cands = [hs[c] for c in range(n) if c != q]
            f, bp, bs, im = oracle.get_close(pred, float(cutoff), hs[q], cands)
            assert np.array_equal(f, v["get_close_flags_%d" % ci][q])
            gbp, gbs, gim = v["get_close_best_%d" % ci][q]
            assert (bp, im) == (int(gbp), bool(gim)) and bs == pytest.approx(gbs, rel=1e-11)
            assert np.array_equal(oracle.filter_(pred, float(cutoff), hs[q], cands), v["filter_%d" % ci][q])
            if q + 1 < n:
                assert oracle.merge(pred, float(cutoff), hs, q, q + 1, min(n - 1, q + 6)) == int(v["merge_%d" % ci][q])
    mem = [hs[i] for i in v["mean_members"]]
    mean, d, near = oracle.mean_nearest(mem)
    # the reference build contracts 1 - frac*frac into an FMA (-march=x86-64-v3); the oracle is built with -ffp-contract=off
    assert np.array_equal(mean, v["mean"]) and np.allclose(d, v["mean_dist"], rtol=1e-12, atol=0) and near == int(v["mean_nearest"])
    c = oracle.Hist()
    oracle.lib().orc_hist_clone(hs[0], c)
    oracle.lib().orc_hist_set(c, hs[min(5, n - 1)])
    assert c.mag == int(v["stale_mag"])
    for f, (name, bit) in enumerate(FEATS):
        assert oracle.raw_feature(1 << bit, c, hs[min(3, n - 1)]) == pytest.approx(v["stale_raw"][f], rel=1e-11, abs=1e-15)


def test_weights_file_round_trip(oracle):
    for name in ("weights_k5_u16.txt", "weights_k9_u32.txt"):
        text = weights_text(name)
        p = oracle.predictor(text)
        again = oracle.predictor_format(p)
        assert again.split() == text.split()      # same tokens (whitespace-insensitive like `in >> tok`)


def test_k13_u64_vectors_printed_by_the_reference(oracle):
    """BASELINE cfg4's parameters (k = 13, uint64_t, 20 kb): vectors_k13_u64.npz holds what the REFERENCE ITSELF computed for six
    such sequences (gen_golden.make_vectors_k13: Loader::get_point, the 11 statistics of predict/Feature.cpp:682-1518, Trainer::get_close /
    filter, get_mean) -- the oracle restatement is held to it here, the GPU in test_cfg4_k13_u64_20kb_against_the_oracle. A subset of the
    pairs (each statistic walks 67 M bins on one core); tolerances: the reference's own sequential FP64 sums of 67 M terms."""
    v = load_vectors("vectors_k13_u64.npz")
    k, dt, n = int(v["k"]), int(v["dtype"]), int(v["n"])
    assert (k, dt, n) == (13, 64, 6)
    seqs, mono = cfg4_sequences()
    assert [bytes(s) for s in v["seqs"]] == [bytes(s) for s in list(seqs) + [mono]]
    oracle.lib().orc_set_threads(8)
    hs = [oracle.hist(bytes(s), k, dt) for s in v["seqs"]]
    for i, h in enumerate(hs):
        a = h.array()
        idx = np.flatnonzero(a != 1)
        assert np.array_equal(idx, v["bins_idx_%d" % i]) and np.array_equal(a[idx], v["bins_val_%d" % i]), i
        assert (h.mag, h.length) == (int(v["mag"][i]), int(v["length"][i])) and list(h.one_mers) == v["one_mers"][i].tolist()
        assert h.stddev == pytest.approx(float(v["stddev"][i]), rel=1e-12)
    assert int(v["bins_val_5"].max()) >= 1 << 16          # the homopolymer run: one count beyond 16 bits
    for i, j in ((0, 1), (1, 0), (0, 3), (5, 0), (4, 1)):
        for f, (name, bit) in enumerate(FEATS):
            val = oracle.raw_feature(1 << bit, hs[i], hs[j])
            if name in EXACT:
                assert val == v["raw"][i, j, f], (name, i, j)
            else:          # same bins, same order of summation, another compiler's contraction of a*b+c: 67 M terms
                assert val == pytest.approx(v["raw"][i, j, f], rel=1e-9, abs=1e-15), (name, i, j)
    pred = oracle.predictor(weights_text("weights_cfg4_k13.txt"))
    for ci, cutoff in enumerate(v["cutoffs"]):
        q = 0
        cands = [hs[c] for c in range(n) if c != q]
        f, bp, bs, im = oracle.get_close(pred, float(cutoff), hs[q], cands)
        gbp, gbs, gim = v["get_close_best_%d" % ci][q]
        assert np.array_equal(f, v["get_close_flags_%d" % ci][q]) and (bp, im) == (int(gbp), bool(gim)) and bs == pytest.approx(gbs, rel=1e-9)
        assert np.array_equal(oracle.filter_(pred, float(cutoff), hs[q], cands), v["filter_%d" % ci][q])
    mean, d, near = oracle.mean_nearest([hs[i] for i in v["mean_members"]])
    ix = np.flatnonzero(mean != 1.0)
    assert np.array_equal(ix, v["mean_idx"]) and np.array_equal(mean[ix], v["mean_val"])
    # distance_d = 10000 (1 - frac^2) with frac = 0.99998: the subtraction leaves 11 digits of the FP64 ratio (and the reference build
    # contracts 1 - frac * frac into one FMA)
    assert np.allclose(d, v["mean_dist"], rtol=1e-9, atol=0) and near == int(v["mean_nearest"])
    for h in hs:
        oracle.lib().orc_hist_free(h)


def test_runs_of_more_than_a_million_bases_are_cut_as_the_reference_cuts_them(oracle):
    """Chromosome::makeSegmentList under help(1000000) (nonltr/Chromosome.cpp:355-385,115-128): floor(len / 1e6) fragments per run of
    unambiguous bases, the last one taking the remainder; k-mers across a cut are not counted. Segments, effective length, histograms
    (k = 5 / 32-bit, k = 7 / 16-bit), magnitude, 1-mers and stddev next to what the reference itself produced (long_fragments.npz)."""
    import os
    v = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "long_fragments.npz"))
    seqs = long_fragment_set()
    assert [len(s) for s in seqs] == v["lengths"].tolist()
    for i, s in enumerate(seqs):
        _, segs, eff = oracle.encode(s)
        assert [list(x) for x in segs] == v["segs_%d" % i].tolist() and eff == int(v["eff_%d" % i])
        for k, dt in ((5, 32), (7, 16)):
            h = oracle.hist(s, k, dt)
            assert np.array_equal(h.array(), v["bins_k%d_%d" % (k, i)]), (i, k)
            meta = v["meta_k%d_%d" % (k, i)]
            assert (h.mag, h.length) == (int(meta[0]), int(meta[1])) and list(h.one_mers) == meta[2:].tolist()
            assert h.stddev == pytest.approx(float(v["stddev_k%d_%d" % (k, i)]), rel=1e-12)
            # the cut is visible in the counts: one fragment boundary = k - 1 k-mers fewer than an uncut run would give
            n_kmers = int(h.array().astype(np.int64).sum()) - 4 ** k
            assert n_kmers == sum(e - b + 1 - (k - 1) for b, e in segs if e - b + 1 >= k)
            oracle.lib().orc_hist_free(h)
    assert len(v["segs_1"]) == 2 and len(v["segs_2"]) == 4          # the cuts really are there
