"""The travelling CPU oracle (oracle/msc_oracle.c) against the committed golden fixtures, which were produced by the
real reference (tests/golden/gen_golden.py). Runs anywhere (no GPU, no /root/reference)."""
import numpy as np
import pytest

from golden_util import EXACT, FEATS, VECTOR_SETS, dense_bins, kat, load_vectors, weights_text


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
