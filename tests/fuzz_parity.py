#!/usr/bin/env python3
"""Randomised parity run: libmeshclust2_hip.so against the CPU oracle over random (k, datatype, layout, lengths, alphabets) -- run on
the GPU box.   python tests/fuzz_parity.py [seconds] [first seed]
Test infrastructure like tests/: the oracle is the checker. Every round builds a random batch in the dense and (k >= 6) the sparse
layout and compares bins, scalar records, all 11 raw statistics for random (candidate, query) pairs in both argument orders, and
get_close / filter / merge / mean_nearest under a fixture model. Integer statistics must be bit-equal, FP64 ones within 1e-9.
Prints one line per round and stops at the first mismatch (exit 1)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import EXACT, FEATS, weights_text
from meshclust2_amd import api
from oracle import oracle_py as orc

MASK = sum(1 << b for _, b in FEATS)
MODELS = ["weights_k5_u16.txt", "weights_k5_u16_slow.txt", "weights_k9_u32.txt", "weights_k8_u16.txt", "weights_mixed_slow_k6_u16.txt"]


def rand_seq(rng):
    kind = rng.integers(0, 10)
    n = int(np.exp(rng.uniform(np.log(25), np.log(6000))))
    if kind == 0:
        p = [0.7, 0.1, 0.1, 0.1]
    elif kind == 1:
        p = [0.45, 0.05, 0.05, 0.45]
    else:
        p = None
    s = bytearray(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n, p=p).tobytes())
    if kind == 2:                                   # a long homopolymer / dinucleotide run: saturation for narrow types
        a = int(rng.integers(0, n))
        run = (b"A" if rng.integers(0, 2) else b"AC") * int(rng.integers(50, 900))
        s[a:a] = run
    if kind in (3, 4):                              # N runs of every length class (merge < 10, split >= 10, short leftovers)
        for _ in range(int(rng.integers(1, 6))):
            a = int(rng.integers(0, len(s)))
            s[a:a + int(rng.integers(1, 40))] = b"N" * int(rng.integers(1, 40))
    if kind == 5:
        for _ in range(int(rng.integers(1, 8))):    # IUPAC ambiguity codes
            s[int(rng.integers(0, len(s)))] = int(rng.choice(np.frombuffer(b"RYMKSWHBVD", dtype=np.uint8)))
    if kind == 6:
        s = bytearray(bytes(s).lower())
    return bytes(s)


class Mismatch(AssertionError):
    pass


def check(cond, what):
    if not cond:
        raise Mismatch(repr(what))


def run_round(ctx, seed):
    """one random configuration through every comparison; raises Mismatch; returns a one-line summary"""
    rng = np.random.default_rng(seed)
    k = int(rng.integers(1, 12))
    dtype = int(rng.choice([8, 16, 32, 64]))
    if k >= 11 and dtype == 64:
        dtype = 32
    n = int(rng.integers(4, 28))
    seqs = [rand_seq(rng) for _ in range(n)]
    base = seqs[0]
    for i in range(1, n, 3):                        # relatives of the first sequence: shared bins, close pairs
        m = bytearray(base)
        for _ in range(max(1, len(m) // 30)):
            m[int(rng.integers(0, len(m)))] = int(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8)))
        seqs[i] = bytes(m)
    oh = [orc.hist(s, k, dtype) for s in seqs]
    layouts = ["dense"] + (["sparse"] if 4 ** k * dtype // 8 >= 65536 else [])      # the sparse layout exists from 64 KiB histograms up
    model_name = MODELS[int(rng.integers(0, len(MODELS)))]
    pred = orc.predictor(weights_text(model_name))
    for layout in layouts:
        hs = api.HistogramSet(ctx, k, dtype, n + 1, sparse_entries=(sum(len(s) for s in seqs) * 2 + 4096) if layout == "sparse" else 0)
        hs.build(seqs)
        for i in range(n):
            inf = hs.info(i)
            check((inf["mag"], inf["length"], inf["one_mers"], inf["overflow"]) == (oh[i].mag, oh[i].length, list(oh[i].one_mers), oh[i].overflow), ("info", seed, layout, i))
            if 4 ** k * dtype // 8 <= (8 << 20):
                check(np.array_equal(hs.download(i), oh[i].array()), ("bins", seed, layout, i))
        live = [i for i in range(n) if oh[i].length > 0]
        if len(live) < 3:
            hs.close()
            continue
        cands = np.array(live, dtype=np.uint32)
        for q in rng.choice(live, size=min(3, len(live)), replace=False):
            for order in (api.ORDER_CAND_FIRST, api.ORDER_QUERY_FIRST):
                raw = api.pair_features_raw(ctx, hs, cands, hs, int(q), MASK, order)
                for ci, c in enumerate(live):
                    a, b = (oh[c], oh[int(q)]) if order == api.ORDER_CAND_FIRST else (oh[int(q)], oh[c])
                    for col, (name, bit) in zip(raw[ci], FEATS):
                        exp = orc.raw_feature(1 << bit, a, b)
                        if name in EXACT and name != "kulczynski2":
                            check(col == exp or (np.isnan(col) and np.isnan(exp)), ("raw", seed, layout, name, c, int(q), order, col, exp))
                        else:
                            check((np.isnan(col) and np.isnan(exp)) or abs(col - exp) <= 1e-9 * max(abs(exp), 1e-4), ("raw", seed, layout, name, c, int(q), order, col, exp))
        feat = api.Feature.from_text(ctx, weights_text(model_name), 0)
        for cutoff in (0.9, 0.6):
            trn = api.Trainer(ctx, feat, cutoff)
            q = int(rng.choice(live))
            w = np.array([c for c in live if c != q], dtype=np.uint32)
            try:
                f1, bp1, bs1, im1 = trn.get_close(hs, w, hs, q)
                gpu_err = None
            except api.MscError as e:
                gpu_err = e
            try:
                f2, bp2, bs2, im2 = orc.get_close(pred, cutoff, oh[q], [oh[c] for c in w])
                cpu_err = None
            except Exception as e:      # noqa: BLE001
                cpu_err = e
            check((gpu_err is None) == (cpu_err is None), ("get_close error parity", seed, layout, gpu_err, cpu_err))
            if gpu_err is None:
                check(np.array_equal(f1, f2) and (bp1, im1) == (bp2, im2) and (bp1 < 0 or abs(bs1 - bs2) <= 1e-9 * max(abs(bs2), 1e-6)), ("get_close", seed, layout, q, cutoff))

            def throws(a, b):          # the reference throws where compute() meets a zero length or a NaN after normalisation
                try:
                    orc.compute(pred.cls, a, b)
                    return False
                except ValueError:
                    return True
            idc = cutoff / 100.0 if cutoff > 1 else cutoff
            lo_len, hi_len = int(oh[q].length * idc), int(oh[q].length / idc)
            window_throws = any(lo_len <= oh[c].length <= hi_len and throws(oh[q], oh[c]) for c in w)
            try:
                kept, ferr = trn.filter(hs, q, hs, w), None
            except api.MscError as e:
                kept, ferr = None, e
            check((ferr is not None) == window_throws, ("filter error parity", seed, layout, q, cutoff, ferr, window_throws))
            if ferr is None:
                check(np.array_equal(kept, orc.filter_(pred, cutoff, oh[q], [oh[c] for c in w])), ("filter", seed, layout, q, cutoff))
            lv = np.array(live, dtype=np.uint32)
            cur = int(rng.integers(0, len(live)))
            last = min(len(live) - 1, cur + 5)
            if cur + 1 <= last:
                want = orc.merge(pred, cutoff, [oh[c] for c in live], cur, cur + 1, last)
                try:
                    got, merr = trn.merge(hs, lv, cur, cur + 1, last), None
                except api.MscError as e:
                    got, merr = None, e
                check((merr is not None) == (want == -2), ("merge error parity", seed, layout, cur, merr, want))
                if merr is None:
                    check(got == want, ("merge", seed, layout, cur, got, want))
        mem = np.array(sorted(set(int(x) for x in rng.choice(live, size=min(6, len(live))))), dtype=np.uint32)
        pos, d, _ = api.mean_nearest(ctx, hs, mem)
        _, od, onear = orc.mean_nearest([oh[int(i)] for i in mem])
        check(pos == onear and np.allclose(d, od, rtol=1e-12, atol=0), ("mean_nearest", seed, layout, mem.tolist()))
        hs.close()
    for h in oh:
        orc.lib().orc_hist_free(h)
    return "seed %d ok: k=%d u%d n=%d layouts=%s model=%s" % (seed, k, dtype, n, "+".join(layouts), model_name)


def run_multi_round(ctx, seed):
    """Q x M pass (msc_score_multi: digest / ring / register / sparse-queued kernels, whichever the library picks) against one
    1 x M pass per query (msc_score, msc_pair_features_raw -- the path run_round pins to the oracle): random k, datatype, layout,
    candidate subset and order, query count, queries from the same or from another set, model with or without the earth
    mover's distance, either argument order. Sums, classify sums and raw statistics must be equal, close flags identical."""
    rng = np.random.default_rng(100000 + seed)
    k = int(rng.integers(3, 11))
    dtype = int(rng.choice([8, 16, 32, 64]))
    n = int(rng.integers(8, 200))
    L = int(np.exp(rng.uniform(np.log(60), np.log(3000))))
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    fam = max(2, int(rng.integers(2, 12)))
    seqs = []
    for i in range(n):
        if i % fam == 0:
            tmpl = bytearray(rng.choice(alpha, max(30, L + int(rng.integers(-L // 5, L // 5 + 1)))).tobytes())
        m = bytearray(tmpl)
        for _ in range(len(m) // 25):
            m[int(rng.integers(0, len(m)))] = int(rng.choice(alpha))
        if rng.integers(0, 20) == 0:
            m[0:0] = b"A" * int(rng.integers(100, 600))          # a run that lifts the largest count (u16-count digest, ring fallbacks)
        seqs.append(bytes(m))
    sparse = bool(rng.integers(0, 4) == 0) and 4 ** k * dtype // 8 >= 65536
    ent = (sum(len(s) for s in seqs) * 2 + 4096) if sparse else 0
    hs = api.HistogramSet(ctx, k, dtype, n, sparse_entries=ent)
    hs.build(seqs)
    other = bool(rng.integers(0, 3) == 0)
    qs = hs
    if other:
        qs = api.HistogramSet(ctx, k, dtype, n, sparse_entries=ent)
        qs.build(seqs[::-1])
    model_name = MODELS[int(rng.integers(0, len(MODELS)))]
    feat = api.Feature.from_text(ctx, weights_text(model_name), 0)
    n_q = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 40]))
    q_slots = rng.integers(0, n, size=n_q).astype(np.uint32)
    cands = rng.permutation(n)[: int(rng.integers(1, n + 1))].astype(np.uint32) if rng.integers(0, 3) else None
    m = n if cands is None else cands.size
    order = api.ORDER_CAND_FIRST if rng.integers(0, 2) else api.ORDER_QUERY_FIRST
    mask = 0
    if rng.integers(0, 2):
        for _, b in FEATS:
            if rng.integers(0, 2):
                mask |= 1 << b
    got = api.score_multi(ctx, feat, hs, cands, qs, q_slots, order=order, m=m, feat_mask=mask)
    kernel = ctx.last_kernel_info()[0]
    for i, q in enumerate(q_slots):
        one = feat.compute(hs, cands, qs, int(q), order=order, m=m)
        check(np.array_equal(got["sum"][i], one["sum"], equal_nan=True) or np.allclose(got["sum"][i], one["sum"], rtol=1e-9, atol=1e-12, equal_nan=True),
              ("multi sum", seed, kernel, k, dtype, i))
        check(np.allclose(got["csum"][i], one["csum"], rtol=1e-9, atol=1e-12, equal_nan=True), ("multi csum", seed, kernel, i))
        # a flag may differ only where the rounding argument sits within 1e-9 of the threshold, which random data does not hit
        check(np.array_equal(got["close"][i], (np.round(one["csum"]) > 0).astype(np.uint8)), ("multi close", seed, kernel, i))
        if mask:
            raw = api.pair_features_raw(ctx, hs, cands, qs, int(q), mask, order, m=m)
            check(np.allclose(got["raw"][i], raw, rtol=1e-9, atol=1e-13, equal_nan=True), ("multi raw", seed, kernel, i))
    hs.close()
    if other:
        qs.close()
    return "multi seed %d ok: k=%d u%d n=%d m=%d q=%d %s%s model=%s kernel=%s" % (seed, k, dtype, n, m, n_q, "sparse " if sparse else "", "other-set " if other else "",
                                                                               model_name, kernel)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ctx = api.Context(0)
    t_end = time.time() + budget
    seed = seed0
    while time.time() < t_end:
        try:
            print(run_round(ctx, seed), flush=True)
            for sub in range(4):
                print(run_multi_round(ctx, 4 * seed + sub), flush=True)
        except Mismatch as e:
            print("MISMATCH:", e, flush=True)
            sys.exit(1)
        seed += 1
    print("fuzz ok: %d rounds, seeds %d..%d" % (seed - seed0, seed0, seed - 1))


if __name__ == "__main__":
    main()
