"""Helper run as a subprocess by test_gpu_parity.py::test_multi_ring_variants (the library reads MSC_MULTI_TQ /
MSC_RING_SLOTS once per process): the LDS-DMA ring form of the Q x M kernel == independent 1 x M passes, bit for bit."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshclust2_amd import api, synth  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from golden_util import FAST  # noqa: E402

FAST_MASK = sum(1 << b for _, b in FAST)


def main():
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    ctx = api.Context(0)
    for dtype, k, n, length in ((32, 9, 330, 1000), (64, 6, 700, 300)):
        seqs, _ = synth.families(900 + k, n, length, family=10)
        hs = api.HistogramSet(ctx, k, dtype, len(seqs))
        hs.build(seqs)
        feat = api.Feature.from_text(ctx, open(os.path.join(golden, "weights_k9_u32.txt")).read(), 0)
        mask = FAST_MASK
        for nq, cands in ((4, np.arange(n, dtype=np.uint32)), (9, np.arange(5, n, dtype=np.uint32)), (16, np.arange(n - 1, -1, -1, dtype=np.uint32)),
                          (8, np.array([7], dtype=np.uint32)), (5, np.array([3, 9, 4], dtype=np.uint32)),
                          (40, np.arange(n, dtype=np.uint32)), (64, np.arange(3, n, dtype=np.uint32)), (33, np.arange(n - 1, 100, -1, dtype=np.uint32))):
            qs = (np.arange(nq, dtype=np.uint32) * 3) % n
            multi = api.score_multi(ctx, feat, hs, cands, hs, qs, feat_mask=mask)
            want_kernel = os.environ.get("MSC_TEST_EXPECT_KERNEL")
            if want_kernel and dtype == 32 and nq == 40:          # the switches under test select the route they say they select
                assert ctx.last_kernel_info()[0].startswith(want_kernel), (ctx.last_kernel_info(), want_kernel)
            only = api.score_multi(ctx, feat, hs, cands, hs, qs, want=("close",))          # flags alone: the f32 screen (pair_features.hip) where the route has it
            assert np.array_equal(only["close"], multi["close"]), (dtype, k, nq)
            for i, q in enumerate(qs):
                single = feat.compute(hs, cands, hs, int(q))
                raw = api.pair_features_raw(ctx, hs, cands, hs, int(q), mask)
                assert np.array_equal(multi["sum"][i], single["sum"]), (dtype, k, nq, i)
                assert np.array_equal(multi["raw"][i], raw), (dtype, k, nq, i)
                assert np.array_equal(multi["close"][i], (np.round(single["csum"]) > 0).astype(np.uint8))
    # counts of 3 .. 8 in most tiles (3 kb sequences over the 16 384 bins of k = 7): the level products beyond the first, tile by tile;
    # more queries than one block of 64, windows shorter than a workgroup's 64 candidates, slot lists
    seqs, _ = synth.families(913, 150, 3000, family=10)
    # (the second round: a 12-mer repeated 11 times in every third sequence -- counts of 9 .. 16, the four-bit levels)
    rep = [s if i % 3 else s[:1500] + (b"ACGTTGCAAGTC" if isinstance(s, bytes) else "ACGTTGCAAGTC") * 11 + s[1500:] for i, s in enumerate(seqs)]
    for dtype, seqs in ((16, seqs), (8, seqs), (32, seqs), (16, rep), (32, rep)):
        hs = api.HistogramSet(ctx, 7, dtype, len(seqs))
        hs.build(seqs)
        top = int(max(hs.download(i).max() for i in (0, 3, 7, 77)))
        assert (9 <= top <= 16) if seqs is rep else (3 <= top <= 8), top
        n = len(seqs)
        for nq, cands in ((2, np.arange(n, dtype=np.uint32)), (65, np.arange(n - 1, -1, -1, dtype=np.uint32)), (130, np.arange(3, 40, dtype=np.uint32)),
                          (7, np.array([5, 5, 9], dtype=np.uint32)), (64, np.arange(0, n, 2, dtype=np.uint32))):
            qs = (np.arange(nq, dtype=np.uint32) * 7) % n
            multi = api.score_multi(ctx, feat, hs, cands, hs, qs, feat_mask=mask, want=("sum", "csum", "close", "counts"))
            assert np.array_equal(multi["counts"], multi["close"].sum(axis=1, dtype=np.uint64)), ("counts", dtype, nq)      # msc_last_close_counts
            for i in range(0, nq, 1 if nq <= 16 else 9):
                raw = api.pair_features_raw(ctx, hs, cands, hs, int(qs[i]), mask)
                single = feat.compute(hs, cands, hs, int(qs[i]))
                assert np.array_equal(multi["raw"][i], raw), ("levels", dtype, nq, i, ctx.last_kernel_info()[0])
                assert np.array_equal(multi["sum"][i], single["sum"]), ("levels", dtype, nq, i)
    # one long sequence (70 kb: more k-mers than a 16-bit prefix of excess counts holds) switches the whole pass to 32-bit prefixes
    seqs, _ = synth.families(977, 40, 1000, family=10)
    long_seqs, _ = synth.families(978, 2, 70000, family=2)
    seqs = list(seqs) + list(long_seqs)
    hs = api.HistogramSet(ctx, 9, 32, len(seqs))
    hs.build(seqs)
    feat = api.Feature.from_text(ctx, open(os.path.join(golden, "weights_k9_u32.txt")).read(), 0)
    qs = np.array([41, 0, 40, 7, 9, 11, 13, 15], dtype=np.uint32)
    multi = api.score_multi(ctx, feat, hs, None, hs, qs, m=len(seqs), feat_mask=FAST_MASK)
    for i, q in enumerate(qs):
        raw = api.pair_features_raw(ctx, hs, None, hs, int(q), FAST_MASK, m=len(seqs))
        assert np.array_equal(multi["raw"][i], raw), ("long", i)
    # counts above 255 (a 400-base homopolymer run) switch the digest kernel to its 16-bit count form; 20 queries = two query
    # groups of 16, the second one partly padded; then slots are overwritten and the next pass must see the new contents
    seqs, _ = synth.families(981, 60, 1000, family=10)
    seqs = list(seqs)
    seqs[7] = seqs[7][:300] + (b"A" if isinstance(seqs[7], bytes) else "A") * 400 + seqs[7][300:]
    hs = api.HistogramSet(ctx, 9, 32, len(seqs))
    hs.build(seqs)
    assert hs.download(7).max() >= 256
    qs = (np.arange(20, dtype=np.uint32) * 7) % len(seqs)
    for rnd in range(2):
        multi = api.score_multi(ctx, feat, hs, None, hs, qs, m=len(seqs), feat_mask=FAST_MASK)
        for i, q in enumerate(qs):
            raw = api.pair_features_raw(ctx, hs, None, hs, int(q), FAST_MASK, m=len(seqs))
            single = feat.compute(hs, None, hs, int(q), m=len(seqs))
            assert np.array_equal(multi["raw"][i], raw), ("wide counts", rnd, i)
            assert np.array_equal(multi["sum"][i], single["sum"]), ("wide counts", rnd, i)
        # overwrite a candidate slot and a query slot (any cached re-encoding of them is stale now)
        hs.build(seqs[20:22], first_slot=3)
        hs.assign_from(int(qs[1]), hs, 30)
    # HBM-resident stage: 6000 x 1 MiB histograms (past L2 and the memory-side cache), every wave runs hundreds of iterations
    n, k, dtype = 6000, 9, 32
    codes = [synth.member(5, t // 20, t % 20, synth.template(5, t // 20, 1000)) for t in range(2000)]
    b = synth.pack_batch(codes)
    hs = api.HistogramSet(ctx, k, dtype, n)
    for done in range(0, n, 2000):
        hs.build_packed(done, 2000, b["packed"], b["n_bases"], b["seg_seq"], b["seg_start"], b["seg_end"], b["eff_len"], b["one_mers"])
    feat = api.Feature.from_text(ctx, open(os.path.join(golden, "weights_k9_u32.txt")).read(), 0)
    qs = np.array([0, 21, 45, 77, 1999, 1033, 512, 5, 300, 1500, 20, 40, 60, 81, 101, 1234], dtype=np.uint32)
    multi = api.score_multi(ctx, feat, hs, None, hs, qs, m=n, want=("sum",))
    for i, q in enumerate(qs):
        single = feat.compute(hs, None, hs, int(q), m=n)
        assert np.array_equal(multi["sum"][i], single["sum"]), ("big", i)
    print("RING_VARIANT_OK")


if __name__ == "__main__":
    main()
