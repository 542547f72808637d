#!/usr/bin/env python3
"""Randomised rounds aimed at the matrix-core route of the Q x M pass (msc_pair_gemm.hip + msc_emd_ranks.hip): sets whose counts stay
small (sequences much shorter than 4^k), random k / bin type / window / slot lists with repeats / query blocks of 2 .. 200 from the same
or another set / both argument orders -- msc_score_multi against one 1 x M pass per query (independent raw-bin kernels): integer
statistics and decisions bit-equal, sums equal.   python tests/fuzz_gemm_route.py [seconds] [first seed]     (run on the GPU box)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import FAST, weights_text  # noqa: E402
from meshclust2_amd import api  # noqa: E402

MASK = sum(1 << b for _, b in FAST)


def one_round(ctx, seed):
    rng = np.random.default_rng(seed)
    k = int(rng.integers(5, 10))
    dtype = int(rng.choice([8, 16, 32]))
    nbins = 4 ** k
    n = int(rng.integers(12, 160))
    top = max(60, nbins // int(rng.choice([8, 16, 64])))
    base = ["".join("ACGT"[b] for b in rng.integers(0, 4, int(rng.integers(40, top)))) for _ in range(max(2, n // 4))]
    seqs = []
    for i in range(n):          # families: copies with a few substitutions, so that some pairs are close
        s = list(base[i % len(base)])
        for _ in range(int(rng.integers(0, 1 + len(s) // 20))):
            s[int(rng.integers(0, len(s)))] = "ACGT"[int(rng.integers(0, 4))]
        if seed % 3 == 0 and i % 4 == 0:          # a short unit repeated: counts up to ~16 (the four-bit levels), sometimes past them (the digest route)
            at = int(rng.integers(0, len(s)))
            s[at:at] = list("".join("ACGT"[b] for b in rng.integers(0, 4, int(rng.integers(k, 2 * k)))) * int(rng.integers(2, 18)))
        seqs.append("".join(s))
    hs = api.HistogramSet(ctx, k, dtype, n)
    hs.build(seqs)
    other = bool(rng.integers(0, 3) == 0)
    if other:
        nq_cap = int(rng.integers(2, 80))
        qset = api.HistogramSet(ctx, k, dtype, nq_cap)
        qset.build([seqs[int(j)] for j in rng.integers(0, n, nq_cap)])
    else:
        qset, nq_cap = hs, n
    n_q = int(rng.integers(2, 200))
    q_slots = rng.integers(0, nq_cap, n_q).astype(np.uint32)
    if rng.integers(0, 2):
        cands, m = rng.integers(0, n, int(rng.integers(1, 2 * n))).astype(np.uint32), None
    else:
        cands, m = None, int(rng.integers(1, n + 1))
    order = int(rng.integers(0, 2))
    text = weights_text("weights_k9_u32.txt").replace("k: 9", "k: %d" % k).replace("uint32_t", "uint%d_t" % dtype)
    feat = api.Feature.from_text(ctx, text, 0)
    got = api.score_multi(ctx, feat, hs, cands, qset, q_slots, order=order, m=m, feat_mask=MASK, want=("sum", "csum", "close", "counts"))
    kernel = ctx.last_kernel_info()[0]
    assert np.array_equal(got["counts"], got["close"].sum(axis=1, dtype=np.uint64)), ("counts", seed, kernel)
    for i in range(0, n_q, 1 if n_q <= 12 else int(rng.integers(3, 17))):
        q = int(q_slots[i])
        raw = api.pair_features_raw(ctx, hs, cands, qset, q, MASK, order=order, m=m)
        one = feat.compute(hs, cands, qset, q, order=order, m=m)
        assert np.array_equal(got["raw"][i], raw), ("raw", seed, kernel, k, dtype, i)
        assert np.array_equal(got["sum"][i], one["sum"]) and np.array_equal(got["csum"][i], one["csum"]), ("sum", seed, kernel, i)
        assert np.array_equal(got["close"][i], (np.round(one["csum"]) > 0).astype(np.uint8)), ("close", seed, kernel, i)
    return "seed %d ok: k=%d u%d n=%d q=%d %s%s kernel=%s" % (seed, k, dtype, n, n_q, "list " if cands is not None else "", "other-set " if other else "", kernel)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ctx = api.Context(0)
    t0, n, gemm = time.time(), 0, 0
    while time.time() - t0 < budget:
        line = one_round(ctx, seed)
        gemm += "k_pair_gemm_fp4_dma" in line
        print(line, flush=True)
        seed += 1
        n += 1
    print("fuzz ok: %d rounds, %d of them through the matrix-core kernel" % (n, gemm))


if __name__ == "__main__":
    main()
