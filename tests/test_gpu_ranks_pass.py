"""The 1 x M pass over rank lists (csrc/msc_ranks_pass.hip: Trainer::get_close / filter shape, cluster/Trainer.cpp:26-61, for histograms of
up to 4^9 bins) held to the CPU oracle's raw statistics (predict/Feature.cpp) and to the merge kernels it replaces -- sparse sets and the
sparse mirrors of dense sets, repeat-bearing sequences (bins that are large in the query, in the candidate, in both), length windows,
slot lists with repeats. The kernel the library ran is asserted by name."""
import os

import numpy as np
import pytest

from golden_util import EXACT, FEATS, weights_text
from meshclust2_amd import api, synth

pytestmark = pytest.mark.gpu
FAST_MASK = sum(1 << b for name, b in FEATS if name not in ("jefferey_divergence", "jensen_shannon"))
RTOL = 1e-9
KERNEL = "k_pair_ranks_1xm"
RANK_KERNELS = ("k_pair_ranks_1xm", "k_pair_ranks_items")          # (the second: lists of more than 8 192 k-mers)


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


@pytest.fixture()
def rank_pass_now(monkeypatch):
    """the library builds a set's rank lists at the third request for the same state of the set; here at the first"""
    monkeypatch.setenv("MSC_RANKS_1XM_AFTER", "1")
    monkeypatch.delenv("MSC_NO_RANKS_1XM", raising=False)
    return monkeypatch


def _sequences(seed, n, length, kind):
    seqs, _ = synth.families(seed, n, length, family=6, length_jitter=length // 10)
    out = []
    for i, s in enumerate(seqs):
        s = bytes(s)
        if kind and i % 5 == 1:          # a run spliced in: bins with large counts, shared by the members that carry the same run
            if kind == "many":          # hundreds of DIFFERENT units, each five to twelve times over: more than 512 bins the query holds three times
                rng = np.random.default_rng(4242)          # and more (its hash table overflows: the look-up falls back to its rank list), hundreds it holds
                run = b"".join(bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 14)]) * int(rng.integers(5, 13)) for _ in range(160))      # >= 8 times
            else:
                run = {"homo": b"A" * 300, "di": b"AC" * 150, "unit3": b"ACG" * 40, "unit12": b"ACGTTGCAAGTC" * 9}[kind]
            at = 50 + 7 * (i % 40)
            s = s[:at] + run + s[at:]
        out.append(s)
    out.append(out[0][: length // 2])                          # a much shorter list than the query's
    out.append(b"ACGT" * 30 + b"N" * 25 + out[1][:200])        # an interrupted one
    out.append(b"A" * 12)                                      # one k-mer, or none
    return out


@pytest.mark.parametrize("dtype,k,n,length,kind,layout", [
    (8, 9, 60, 1000, None, "sparse"),
    (8, 9, 60, 1000, "homo", "sparse"),          # saturating 8-bit bins
    (16, 9, 60, 1000, "di", "sparse"),
    (32, 9, 60, 1000, "unit12", "dense"),        # the sparse mirror of a dense set
    (32, 9, 60, 1000, "homo", "dense"),
    (16, 8, 50, 2500, "unit3", "sparse"),        # 65 536 bins, lists of 2 500
    (32, 7, 50, 400, None, "dense"),
    (16, 8, 40, 600, "unit3", "dense"),
    (16, 9, 24, 12000, "unit12", "sparse"),      # queries of more than 8 192 k-mers: their rank lists stay in global memory
    (16, 9, 30, 3000, "many", "sparse"),          # more than 512 bins the query holds three times and more: no hash table, its rank list searched
])
def test_rank_pass_against_the_oracle_and_the_merge_kernels(ctx, oracle, rank_pass_now, dtype, k, n, length, kind, layout):
    seqs = _sequences(7000 + 13 * k + dtype, n, length, kind)
    n = len(seqs)
    if layout == "sparse":
        hs = api.HistogramSet(ctx, k, dtype, n, sparse_entries=sum(len(s) for s in seqs) * 2 + 1000)
    else:
        hs = api.HistogramSet(ctx, k, dtype, n)
    hs.build(seqs)
    rng = np.random.default_rng(k * 100 + dtype)
    cands = np.concatenate([np.arange(n, dtype=np.uint32)[::-1], rng.integers(0, n, 7).astype(np.uint32)])
    oh = [oracle.hist(s, k, dtype) for s in seqs]
    fast = [(name, b) for name, b in FEATS if (1 << b) & FAST_MASK]
    for q in (1, 0, 6, n - 3, n - 1):          # slot 1, 6: repeat-bearing queries; n - 3: half a sequence; n - 1: (almost) empty
        for order in (api.ORDER_CAND_FIRST, api.ORDER_QUERY_FIRST):
            got = api.pair_features_raw(ctx, hs, cands, hs, q, FAST_MASK, order)
            kernel = ctx.last_kernel_info()[0]
            if 4 ** k >= 16384 and dtype != 64:          # (smaller histograms have no list form: the dense kernels take them)
                assert kernel == ("k_pair_ranks_items" if hs.entries(q) > 2000 else KERNEL), (kernel, dtype, k, layout, q)          # (the query's own size decides)
            rank_pass_now.setenv("MSC_NO_RANKS_1XM", "1")
            ref = api.pair_features_raw(ctx, hs, cands, hs, q, FAST_MASK, order)
            assert ctx.last_kernel_info()[0] not in RANK_KERNELS
            rank_pass_now.delenv("MSC_NO_RANKS_1XM")
            assert np.array_equal(got, ref), (q, order, kernel)          # the same integer records -> the same doubles, bit for bit
            for ci in list(range(0, len(cands), 9)) + [len(cands) - 2]:
                c = int(cands[ci])
                a, b = (oh[c], oh[q]) if order == api.ORDER_CAND_FIRST else (oh[q], oh[c])
                for col, (name, bit) in enumerate(fast):
                    exp = oracle.raw_feature(1 << bit, a, b)
                    if name in EXACT and name != "kulczynski2":
                        assert got[ci][col] == exp, (name, q, c, order, kernel)
                    else:
                        assert got[ci][col] == pytest.approx(exp, rel=RTOL, abs=1e-13), (name, q, c, order, kernel)
    for h in oh:
        oracle.lib().orc_hist_free(h)


@pytest.mark.parametrize("dtype,k,sparse", [(8, 9, True), (32, 9, False), (16, 8, True)])
def test_get_close_and_filter_through_the_rank_pass(ctx, rank_pass_now, dtype, k, sparse):
    """Trainer::get_close / filter with their length windows (cluster/Trainer.cpp:39-40): decisions, best candidate and its similarity equal
    to the merge kernels'; a set that is written between two passes falls back until it has been asked for again."""
    seqs = _sequences(8100 + k, 70, 1000, "di")
    n = len(seqs)
    hs = api.HistogramSet(ctx, k, dtype, n + 1, sparse_entries=sum(len(s) for s in seqs) * 2 + 5000) if sparse else api.HistogramSet(ctx, k, dtype, n + 1)
    hs.build(seqs)
    text = weights_text("weights_k9_u32.txt").replace("k: 9", "k: %d" % k).replace("uint32_t", "uint%d_t" % dtype)
    feat = api.Feature.from_text(ctx, text, 0)
    for cutoff in (0.9, 0.6):
        tr = api.Trainer(ctx, feat, cutoff)
        for q in (0, 1, 6, n - 3):
            w = np.array([c for c in range(n) if c != q], dtype=np.uint32)
            a = tr.get_close(hs, w, hs, q)
            assert ctx.last_kernel_info()[0] == KERNEL
            fa = tr.filter(hs, q, hs, w)
            rank_pass_now.setenv("MSC_NO_RANKS_1XM", "1")
            b = tr.get_close(hs, w, hs, q)
            assert ctx.last_kernel_info()[0] != KERNEL
            fb = tr.filter(hs, q, hs, w)
            rank_pass_now.delenv("MSC_NO_RANKS_1XM")
            assert np.array_equal(a[0], b[0]) and a[1:] == b[1:], (q, cutoff)
            assert np.array_equal(fa, fb)
    if sparse:          # the Q x M shape on sparse sets: one rank pass per query, queued back to back (msc_score_multi)
        cands = np.arange(n, dtype=np.uint32)[::-1].copy()
        qs = np.array([0, 1, 6, n - 3, n - 1], dtype=np.uint32)
        a = api.score_multi(ctx, feat, hs, cands, hs, qs, want=("sum", "csum", "close", "counts"))
        assert ctx.last_kernel_info()[0] == KERNEL
        rank_pass_now.setenv("MSC_NO_RANKS_1XM", "1")
        b = api.score_multi(ctx, feat, hs, cands, hs, qs, want=("sum", "csum", "close", "counts"))
        assert ctx.last_kernel_info()[0] != KERNEL
        rank_pass_now.delenv("MSC_NO_RANKS_1XM")
        for key in ("sum", "csum", "close", "counts"):
            assert np.array_equal(a[key], b[key]), key
    # a write makes the lists stale: the next pass runs on the merge kernel (MSC_RANKS_1XM_AFTER=2: not yet asked for twice), the one
    # after it on fresh rank lists -- same answers throughout
    rank_pass_now.setenv("MSC_RANKS_1XM_AFTER", "2")
    tr = api.Trainer(ctx, feat, 0.9)
    w = np.arange(1, n + 1, dtype=np.uint32)
    hs.clone_from(n, hs, 3)
    first = tr.get_close(hs, w, hs, 0)
    k1 = ctx.last_kernel_info()[0]
    second = tr.get_close(hs, w, hs, 0)
    k2 = ctx.last_kernel_info()[0]
    assert k1 != KERNEL and k2 == KERNEL, (k1, k2)
    assert np.array_equal(first[0], second[0]) and first[1:] == second[1:]
    assert first[0][n - 1] == first[0][2]          # (slot n is a copy of slot 3 = window index 2)


@pytest.mark.parametrize("seed", [3, 4, 5, 6, 7, 8, 9, 10])
def test_rank_pass_randomised(ctx, rank_pass_now, seed):
    """Seeded random sets (k, bin type, layout, lengths, repeat units, slot lists with repeats and gaps, cut-offs): get_close decisions and
    the model's sums through the rank pass equal to the merge kernels' -- bit for bit."""
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.choice([8, 9, 9]))
    dtype = int(rng.choice([8, 16, 32]))
    sparse = bool(rng.integers(0, 2)) or dtype == 8 and k == 8          # (a dense 8-bit set at k = 8 is 64 KiB: the smallest with a list form)
    length = int(rng.choice([300, 1000, 2200]))
    n = int(rng.integers(30, 90))
    seqs, _ = synth.families(5000 + seed, n, length, family=int(rng.integers(2, 9)), length_jitter=length // int(rng.choice([5, 10, 40])))
    seqs = [bytes(s) for s in seqs]
    units = [b"A", b"AC", b"ACG", b"ACGTTGCAAGTC", b"GGGGGGGGGT"]
    for i in rng.choice(n, size=n // 4, replace=False):
        u = units[int(rng.integers(0, len(units)))]
        run = u * int(rng.integers(2, 60))
        run = run[: max(k, min(len(run), 250 if dtype == 8 else 600))]
        at = int(rng.integers(0, len(seqs[i])))
        seqs[i] = seqs[i][:at] + run + seqs[i][at:]
    hs = api.HistogramSet(ctx, k, dtype, n, sparse_entries=sum(len(s) for s in seqs) * 2 + 1000) if sparse else api.HistogramSet(ctx, k, dtype, n)
    hs.build(seqs)
    text = weights_text("weights_k9_u32.txt").replace("k: 9", "k: %d" % k).replace("uint32_t", "uint%d_t" % dtype)
    feat = api.Feature.from_text(ctx, text, 0)
    for trial in range(6):
        q = int(rng.integers(0, n))
        w = rng.integers(0, n, int(rng.integers(1, 2 * n))).astype(np.uint32)
        tr = api.Trainer(ctx, feat, float(rng.choice([0.95, 0.9, 0.7, 0.5])))
        a = tr.get_close(hs, w, hs, q)
        ka = ctx.last_kernel_info()[0]
        sa = feat.compute(hs, w, hs, q)
        rank_pass_now.setenv("MSC_NO_RANKS_1XM", "1")
        b = tr.get_close(hs, w, hs, q)
        kb = ctx.last_kernel_info()[0]
        sb = feat.compute(hs, w, hs, q)
        rank_pass_now.delenv("MSC_NO_RANKS_1XM")
        assert ka in RANK_KERNELS and kb not in RANK_KERNELS, (ka, kb, k, dtype, sparse)
        assert np.array_equal(a[0], b[0]) and a[1:] == b[1:], (seed, trial, k, dtype, sparse)
        assert np.array_equal(sa["sum"], sb["sum"]) and np.array_equal(sa["csum"], sb["csum"]), (seed, trial)


ALL_MASK = sum(1 << b for _, b in FEATS)


@pytest.mark.parametrize("dtype,k,n,length,kind,layout", [
    (16, 9, 50, 1000, None, "sparse"),
    (16, 9, 50, 1000, "homo", "sparse"),          # counts of ~290 in the query and in candidates: cells beyond the table, evaluated on the spot
    (8, 9, 50, 1000, "di", "sparse"),
    (32, 9, 40, 1000, "unit12", "dense"),
    (16, 9, 24, 6000, "unit3", "sparse"),         # lists of 6 000
    (16, 8, 40, 2500, "unit12", "sparse"),
    (16, 9, 20, 12000, "homo", "sparse"),         # long lists: (candidate, round) items; runs of 300 copies across round boundaries
    (32, 9, 20, 10500, "unit3", "dense"),
    (16, 9, 30, 3000, "many", "sparse"),          # ~1 900 bins of count >= 4 in query and candidates: hash overflow, hundreds of spot terms per item (the queue is emptied on the way), a
                                                  # sorted list of the query's large counts longer than a wave
])
def test_divergence_statistics_through_the_rank_pass(ctx, oracle, rank_pass_now, dtype, k, n, length, kind, layout):
    """jefferey_divergence / jensen_shannon (predict/Feature.cpp:1235-1262, 988-1008) counted per cell by k_pair_ranks_items and evaluated by
    k_rank_items_finish: next to the oracle (1e-9) and to the merge kernel (same terms, another order of addition: 1e-12); every other
    statistic of the same pass bit-equal."""
    rank_pass_now.setenv("MSC_RANKS_DIV", "1")          # (outside msc_get_close_window the divergence form is opt-in: DESIGN.md 4.1d)
    seqs = _sequences(9000 + 17 * k + dtype, n, length, kind)
    n = len(seqs)
    hs = api.HistogramSet(ctx, k, dtype, n, sparse_entries=sum(len(s) for s in seqs) * 2 + 1000) if layout == "sparse" else api.HistogramSet(ctx, k, dtype, n)
    hs.build(seqs)
    cands = np.arange(n, dtype=np.uint32)[::-1].copy()
    oh = [oracle.hist(s, k, dtype) for s in seqs]
    div_cols = [c for c, (name, _) in enumerate(FEATS) if name in ("jefferey_divergence", "jensen_shannon")]
    for q in (1, 0, 6, n - 3):
        for order in (api.ORDER_CAND_FIRST, api.ORDER_QUERY_FIRST):
            got = api.pair_features_raw(ctx, hs, cands, hs, q, ALL_MASK, order)
            assert ctx.last_kernel_info()[0] == "k_pair_ranks_items", q          # (r05: a pass with the divergence statistics takes the items kernel whatever its length)
            rank_pass_now.setenv("MSC_NO_RANKS_DIV", "1")
            ref = api.pair_features_raw(ctx, hs, cands, hs, q, ALL_MASK, order)
            assert ctx.last_kernel_info()[0] not in RANK_KERNELS
            rank_pass_now.delenv("MSC_NO_RANKS_DIV")
            for c, (name, bit) in enumerate(FEATS):
                if c in div_cols:
                    assert np.allclose(got[:, c], ref[:, c], rtol=1e-12, atol=1e-15), (name, q, order)
                else:
                    assert np.array_equal(got[:, c], ref[:, c]), (name, q, order)
            for ci in range(0, n, 7):
                cc = int(cands[ci])
                a, b = (oh[cc], oh[q]) if order == api.ORDER_CAND_FIRST else (oh[q], oh[cc])
                for c in div_cols:
                    exp = oracle.raw_feature(1 << FEATS[c][1], a, b)
                    assert got[ci][c] == pytest.approx(exp, rel=RTOL, abs=1e-13), (FEATS[c][0], q, cc, order)
    for h in oh:
        oracle.lib().orc_hist_free(h)


def test_long_outlier_does_not_size_every_pass(ctx, rank_pass_now):
    """One long scaffold among mixed-length sequences: the (candidate, round) pass sizes its accumulators by what the pass's
    length window can meet, not by the set's longest list -- the window path (Trainer::get_close over the accumulate loop's window,
    cluster/ClusterFactory.cpp:553-610) with the `--feat slow` model gives the merge kernels' decisions with the outlier inside the window's
    candidate list (dropped by its length) and as the query itself."""
    rng = np.random.default_rng(77)
    seqs, _ = synth.families(31337, 60, 6000, family=5, length_jitter=2500)
    seqs = [bytes(s) for s in seqs]
    seqs[17] = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 400000)])          # 400 kb: 391 rounds of 1 024
    n = len(seqs)
    hs = api.HistogramSet(ctx, 9, 16, n, sparse_entries=sum(len(s) for s in seqs) + 4096)
    hs.build(seqs)
    feat = api.Feature.from_text(ctx, weights_text("weights_cfg5_k9.txt").replace("uint8_t", "uint16_t"), 0)
    tr = api.Trainer(ctx, feat, 0.6)
    order = np.argsort([len(s) for s in seqs], kind="stable").astype(np.uint32)
    win = api.Window(ctx, hs, order)
    for q in (3, 40, 17):
        alive = np.array([p for p in range(n) if win.alive(p, p + 1)], dtype=np.int64)          # positions of the window before the pass
        close, bp, bs, im = win.get_close(tr, 0, n, hs, q)
        kernel = ctx.last_kernel_info()[0]
        assert kernel == "k_pair_ranks_items", kernel
        rank_pass_now.setenv("MSC_NO_RANKS_1XM", "1")
        ref = tr.get_close(hs, order[alive], hs, q)
        rank_pass_now.delenv("MSC_NO_RANKS_1XM")
        assert ctx.last_kernel_info()[0] not in RANK_KERNELS
        assert np.array_equal(np.sort(close), alive[np.flatnonzero(ref[0])]), q
        assert im == ref[3] and (ref[1] < 0 or (bp == int(alive[ref[1]]) and bs == pytest.approx(ref[2], rel=1e-9)))
