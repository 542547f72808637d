#!/usr/bin/env python3
"""Randomised training rounds against the REAL reference's BestFirstSelector::train_class (through oracle/_ref/libmsc_ref.so's
harness) -- run on the GPU box.   python tests/fuzz_training_vs_reference.py [seconds] [first seed]
Per round: random k, histogram type, feature set on offer (fast / slow), min / max model size, identity threshold and a random
labelled pair set (templates + graded mutants) -> the reference selects and fits on the CPU -> msc_train_class does on the GPU
(feature table) + host (selection, normal equations). Same combos and singles in the same order, bounds within 1e-9, identical
accuracies, weights within 1e-6 unless the fit is ill-conditioned (cond(A^T A) > 1e6: coefficients reported, not compared)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import parse_class_block, training_set
from meshclust2_amd import api
from oracle import ref_py


def close(a, b, rel, abs_=0.0):
    return abs(a - b) <= max(rel * max(abs(a), abs(b)), abs_)


def run_round(ctx, seed):
    rng = np.random.default_rng(seed)
    k = int(rng.integers(3, 9))
    dtype = int(rng.choice([8, 16, 32]))
    slow = bool(rng.integers(0, 3) == 0)
    feat_flags = api.FEAT_SLOW if slow else api.FEAT_FAST
    min_feat = int(rng.integers(1, 4))
    max_feat = int(rng.integers(min_feat, 5))
    ident = float(rng.choice([0.6, 0.7, 0.8, 0.9, 0.95]))
    n_templates, per_template = int(rng.integers(8, 30)), int(rng.integers(4, 12))
    length = int(rng.integers(150, 1200))
    seqs, pairs = training_set(int(rng.integers(1, 1 << 30)), n_templates, per_template, length)
    n_train = len(pairs) // 2
    ref_py.lib().ref_set_threads(1)          # the canonical order of the open list (predict/BestFirstSelector.cpp:156-170)
    pts_ref = [ref_py.Point(dtype, s, k) for s in seqs]
    first, second, vals = [a for a, b, v in pairs], [b for a, b, v in pairs], [v for a, b, v in pairs]
    t0 = time.time()
    etext, eatr, eate = ref_py.train_class(dtype, k, [pts_ref[a] for a in first], [pts_ref[b] for b in second], vals, n_train, feat_flags, min_feat, max_feat, ident)
    t_ref = time.time() - t0
    pts = api.HistogramSet(ctx, k, dtype, len(seqs))
    pts.build(seqs)
    text, atr, ate = api.train_class(ctx, pts, first, second, vals, n_train, feat_flags, min_feat, max_feat, ident)
    w0, combos, singles = parse_class_block(text)
    ew0, ecombos, esingles = parse_class_block(etext)
    what = "seed %d (k=%d u%d %s min=%d max=%d id=%.2f, %d pairs)" % (seed, k, dtype, "slow" if slow else "fast", min_feat, max_feat, ident, len(pairs))
    if ([(c, f) for c, f, _ in combos] != [(c, f) for c, f, _ in ecombos] or [f for f, _, _ in singles] != [f for f, _, _ in esingles]) and \
            close(atr, eatr, 0, 1e-9) and close(ate, eate, 0, 1e-9):
        # The open list is a std::priority_queue keyed by accuracy alone: among candidate sets of EQUAL accuracy the heap's push history
        # decides, and one training sample whose score sits within rounding of the threshold under some intermediate (often
        # ill-conditioned) fit changes that history. Both searches then end on models of the same training and testing accuracy.
        pts.close()
        return "train %s tied: another model of the same accuracy %.1f / %.1f selected (%s vs the reference's %s)" % (
            what, atr, ate, [f for _, f, _ in combos], [f for _, f, _ in ecombos])
    def cond_of(model_text):
        f_ = api.Feature.from_text(ctx, model_text, 0)
        rows_ = []
        trf, trs = first[:n_train], second[:n_train]
        for b in sorted(set(trs)):
            fs = np.array([a for a, bb in zip(trf, trs) if bb == b], dtype=np.uint32)
            rows_.append(f_.compute(pts, fs, pts, b)["combos"])
        A_ = np.hstack([np.ones((n_train, 1)), np.vstack(rows_)])
        return float(np.linalg.cond(A_.T @ A_))
    if [(c, f) for c, f, _ in combos] != [(c, f) for c, f, _ in ecombos] or [f for f, _, _ in singles] != [f for f, _, _ in esingles]:
        # Another set AND another accuracy. One known cause: a candidate set with near-collinear combos (weights like +110 / -96) whose
        # Gauss-Jordan fit is rounding-driven (cond(A^T A) past 1e6) scores another accuracy under the 1e-12 noise of the feature table
        # (seed 8397), and the search follows whichever side ranks it higher. Reported as its own class, not as agreement.
        c_gpu = cond_of(text)
        if c_gpu >= 1e6:
            pts.close()
            return "train %s ILL-CONDITIONED: the search diverged at a candidate set with cond(A^T A) = %.1e (%s, accuracy %.1f / %.1f; the reference chose %s, %.1f / %.1f)" % (
                what, c_gpu, [f for _, f, _ in combos], atr, ate, [f for _, f, _ in ecombos], eatr, eate)
        raise AssertionError("%s: another model selected\nreference (accuracy %r / %r):\n%s\nGPU (accuracy %r / %r):\n%s" % (what, eatr, eate, etext, atr, ate, text))
    w_ok = close(w0, ew0, 1e-6, 1e-9) and all(close(w, ew, 1e-6, 1e-9) for (_, _, w), (_, _, ew) in zip(combos, ecombos))
    note = ""
    if not w_ok:
        # Near-collinear combos (weights like +123 / -123) make the normal equations ill-conditioned, and the reference solves them by
        # Gauss-Jordan without pivoting on A^T A: last-bit differences in the feature table (pearson, kulczynski2, ... are FP64 loops
        # over the bins there, closed forms of integer moments here: 1e-12 apart) move the coefficients in the 3rd..5th digit once
        # cond(A^T A) passes 1e6 (5e-5 at 1e7, 20 % at 2e9 in r01). Then only structure, bounds and accuracies are compared.
        f_gpu = api.Feature.from_text(ctx, text, 0)
        rows = []
        tr_first, tr_second = first[:n_train], second[:n_train]
        for b in sorted(set(tr_second)):
            fs = np.array([a for a, bb in zip(tr_first, tr_second) if bb == b], dtype=np.uint32)
            rows.append(f_gpu.compute(pts, fs, pts, b)["combos"])
        A = np.hstack([np.ones((n_train, 1)), np.vstack(rows)])
        cond = float(np.linalg.cond(A.T @ A))
        wmax = max(abs(ew0), max(abs(ew) for _, _, ew in ecombos))
        worst = max([abs(w0 - ew0)] + [abs(w - ew) for (_, _, w), (_, _, ew) in zip(combos, ecombos)]) / wmax
        # Gauss-Jordan without pivoting on A^T A: past 1e6 the reference's own coefficients are rounding-driven; below it the 1e-12
        # noise of the feature table may show up amplified by the condition number (seed 7322, `slow`: cond 5.5e5, 1.7e-6)
        w_ok = cond >= 1e6 or worst <= cond * 1e-11
        note = " [ill-conditioned fit: cond(A^T A) = %.1e, coefficients differ by %.1e of the largest]" % (cond, worst)
    pts.close()
    ok = w_ok and all(close(lo, elo, 1e-9, 1e-12) and close(hi, ehi, 1e-9, 1e-12) for (_, lo, hi), (_, elo, ehi) in zip(singles, esingles))
    ok = ok and close(atr, eatr, 0, 1e-9) and close(ate, eate, 0, 1e-9)
    if not ok:
        raise AssertionError("%s: same structure, other numbers%s\nreference:\n%s\nacc %r %r\nGPU:\n%s\nacc %r %r" % (what, note, etext, eatr, eate, text, atr, ate))
    return "train %s ok: %d combos %s, accuracy %.1f / %.1f (reference %.1f s)%s" % (what, len(combos), [f for _, f, _ in combos], atr, ate, t_ref, note)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    if not ref_py.available():
        raise SystemExit("oracle/_ref/libmsc_ref.so is not built (needs /root/reference at build time)")
    ctx = api.Context(0)
    t_end = time.time() + budget
    n = ill = 0
    while time.time() < t_end:
        msg = run_round(ctx, seed)
        print(msg, flush=True)
        ill += "ILL-CONDITIONED" in msg
        seed += 1
        n += 1
    print("training fuzz done: %d rounds, %d ILL-CONDITIONED" % (n, ill))
    # the ILL-CONDITIONED class is a statement about the FINAL model, not about the expansion where the two searches parted: a selection
    # bug that happens to end on an ill-conditioned set would hide in it. Its rate is therefore bounded: 2 of 616 rounds so far (r01-r02);
    # more than 3 % of a run (and more than two rounds) fails the fuzz.
    if ill > 2 and ill > 0.03 * n:
        raise SystemExit("too many ILL-CONDITIONED rounds: %d of %d" % (ill, n))


if __name__ == "__main__":
    main()
