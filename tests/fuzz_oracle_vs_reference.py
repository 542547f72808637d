#!/usr/bin/env python3
"""Randomised pinning of the travelling C oracle (oracle/msc_oracle.c) against the REAL reference (oracle/_ref/libmsc_ref.so) -- CPU only,
runs in this container.   python tests/fuzz_oracle_vs_reference.py [seconds] [first seed]
Random k (1..9), histogram type, skewed / repetitive / N-ridden / IUPAC / lower-case sequences: bins, scalar records, the 11 raw
statistics (integer-accumulated ones bitwise, the FP64 loops to 1e-11), get_close / filter / merge under the fixture models and the
mean + nearest member."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as orc
from oracle import ref_py as ref

GOLDEN = os.path.join(ROOT, "tests", "golden")
MODELS = ["weights_k5_u16.txt", "weights_k5_u16_slow.txt", "weights_k9_u32.txt", "weights_k8_u16.txt", "weights_mixed_slow_k6_u16.txt"]
EXACT = ("manhattan", "euclidean", "normalized_vectors", "intersection", "emd", "length_difference", "kulczynski2", "simratio")


def rand_seq(rng):
    kind = rng.integers(0, 10)
    n = int(np.exp(rng.uniform(np.log(25), np.log(3000))))
    p = [0.7, 0.1, 0.1, 0.1] if kind == 0 else [0.45, 0.05, 0.05, 0.45] if kind == 1 else None
    s = bytearray(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n, p=p).tobytes())
    if kind == 2:
        a = int(rng.integers(0, n))
        s[a:a] = (b"A" if rng.integers(0, 2) else b"AC") * int(rng.integers(50, 900))
    if kind in (3, 4):
        for _ in range(int(rng.integers(1, 6))):
            a = int(rng.integers(0, len(s)))
            s[a:a + int(rng.integers(1, 40))] = b"N" * int(rng.integers(1, 40))
    if kind == 5:
        for _ in range(int(rng.integers(1, 8))):
            s[int(rng.integers(0, len(s)))] = int(rng.choice(np.frombuffer(b"RYMKSWHBVD", dtype=np.uint8)))
    if kind == 6:
        s = bytearray(bytes(s).lower())
    return bytes(s)


def check(cond, what):
    if not cond:
        raise AssertionError(repr(what))


def run_round(seed):
    rng = np.random.default_rng(seed)
    k = int(rng.integers(1, 10))
    dtype = int(rng.choice([8, 16, 32, 64]))
    n = int(rng.integers(4, 16))
    seqs = [rand_seq(rng) for _ in range(n)]
    for i in range(1, n, 3):
        m = bytearray(seqs[0])
        for _ in range(max(1, len(m) // 30)):
            m[int(rng.integers(0, len(m)))] = int(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8)))
        seqs[i] = bytes(m)
    oh = [orc.hist(s, k, dtype) for s in seqs]
    rp = [ref.Point(dtype, s, k) for s in seqs]
    for i in range(n):
        check(np.array_equal(oh[i].array(), rp[i].bins()), ("bins", seed, i))
        m = rp[i].meta()
        check((oh[i].mag, oh[i].length, list(oh[i].one_mers)) == (m["mag"], m["length"], m["one_mers"]), ("meta", seed, i))
    # a histogram without a single k-mer is all pseudocounts: pearson is then NaN and the reference TERMINATES inside its OpenMP region
    # (predict/Feature.cpp:143-146) -- the error path is pinned by tests/test_oracle_vs_ref.py through the serial entry points
    live = [i for i in range(n) if oh[i].length > 0 and oh[i].mag > oh[i].nbins]
    if len(live) < 3:
        return "oracle seed %d ok (k=%d u%d, too few usable sequences for the operators)" % (seed, k, dtype)
    for i in live[:6]:
        for j in live[:6]:
            for name in ("manhattan", "euclidean", "normalized_vectors", "jefferey_divergence", "pearson", "intersection", "emd", "length_difference", "kulczynski2",
                         "simratio", "jensen_shannon"):
                a = orc.raw_feature(orc.FEAT[name], oh[i], oh[j])
                b = ref.raw_feature(orc.FEAT[name], rp[i], rp[j])
                if name in EXACT:
                    check(a == b or (np.isnan(a) and np.isnan(b)), ("raw", seed, name, i, j, a, b))
                else:
                    check((np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-11 * max(abs(b), 1e-4), ("raw", seed, name, i, j, a, b))
    model = MODELS[int(rng.integers(0, len(MODELS)))]
    pred = orc.predictor(open(os.path.join(GOLDEN, model)).read())
    rm = ref.Model(dtype, os.path.join(GOLDEN, model))
    for cutoff in (0.9, 0.6):
        q = int(rng.choice(live))
        w = [c for c in live if c != q]
        try:
            of, obp, obs, omin = orc.get_close(pred, cutoff, oh[q], [oh[c] for c in w])
            oerr = None
        except Exception as e:      # noqa: BLE001
            oerr = e
        if oerr is not None or any(orc.merge(pred, cutoff, [oh[c] for c in live], c0, c0 + 1, min(len(live) - 1, c0 + 5)) == -2 for c0 in range(len(live) - 1)):
            # where the oracle reports that the reference throws (zero length, NaN after normalisation), the reference in fact TERMINATES the
            # process from inside its OpenMP region: such windows cannot be replayed through the harness
            continue
        rf, rbp, rbs, rmin = rm.get_close(cutoff, rp[q], [rp[c] for c in w])
        if oerr is None:
            check(np.array_equal(of, rf) and obp == rbp and omin == rmin and (obp < 0 or abs(obs - rbs) <= 1e-11 * max(abs(rbs), 1e-6)), ("get_close", seed, q, cutoff))
            check(np.array_equal(orc.filter_(pred, cutoff, oh[q], [oh[c] for c in w]), rm.filter(cutoff, rp[q], [rp[c] for c in w])), ("filter", seed, q, cutoff))
            cur = int(rng.integers(0, len(live)))
            last = min(len(live) - 1, cur + 5)
            if cur + 1 <= last:
                check(orc.merge(pred, cutoff, [oh[c] for c in live], cur, cur + 1, last) == rm.merge(cutoff, [rp[c] for c in live], cur, cur + 1, last), ("merge", seed, cur))
    mem = sorted(set(int(x) for x in rng.choice(live, size=min(6, len(live)))))
    om, od, onear = orc.mean_nearest([oh[i] for i in mem])
    rmn, rd, rnear = ref.mean_nearest([rp[i] for i in mem])
    # 10000 * (1 - (dist / mag)^2) cancels for members next to the mean: the last bit of the ratio (the reference build contracts to FMA,
    # the oracle build does not) shows as ~1e-12 absolute
    check(np.array_equal(om, rmn) and np.allclose(od, rd, rtol=1e-12, atol=1e-11) and onear == rnear, ("mean_nearest", seed, mem))
    for h in oh:
        orc.lib().orc_hist_free(h)
    return "oracle seed %d ok: k=%d u%d n=%d model=%s" % (seed, k, dtype, n, model)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    if not ref.available():
        raise SystemExit("oracle/_ref/libmsc_ref.so is not built (needs /root/reference at build time)")
    ref.lib().ref_set_threads(1)          # the reference's arg-max among equal maxima follows its thread schedule: one thread = window order
    t_end = time.time() + budget
    n = 0
    while time.time() < t_end:
        print(run_round(seed), flush=True)
        seed += 1
        n += 1
    print("oracle fuzz ok: %d rounds" % n)


if __name__ == "__main__":
    main()
