#!/usr/bin/env python3
"""Randomised driver rounds -- run on the GPU box.   python tests/fuzz_driver.py [seconds] [first seed]
msc_cluster on a random FASTA file (families of mutated templates, mixed lengths, a few N runs) with a fixture model, three ways:
batched update stage on the dense layout, --serial-update (centre by centre), and --sparse. The three .clstr files must be the
same bytes: the batched entry points (msc_update_centres, msc_merge_all, msc_hist_assign_batch), the per-centre path and the
sparse kernels all feed the same clustering logic."""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "meshclust2_amd", "host", "msc_cluster")
CASES = [("weights_k5_u16.txt", 5, 16, 0.9), ("weights_k8_u16.txt", 8, 16, 0.9), ("weights_k9_u32.txt", 9, 32, 0.9), ("weights_k9_u8.txt", 9, 8, 0.9),
         ("weights_mixed_slow_k6_u16.txt", 6, 16, 0.6), ("weights_k5_u16_slow.txt", 5, 16, 0.8),
         # `--feat slow` models where the list form exists: the batched update stage takes its divergence sums from a pair-list pass
         ("weights_cfg5_u16_k9.txt", 9, 16, 0.6), ("weights_cfg5_k9.txt", 9, 8, 0.6), ("weights_cfg5_u16_k9.txt", 9, 16, 0.8)]


def run_round(seed, tmp):
    rng = np.random.default_rng(seed)
    wts, k, dtype, ident = CASES[int(rng.integers(0, len(CASES)))]
    n = int(rng.integers(40, 700))
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    fam = int(rng.integers(2, 25))
    rate = float(rng.choice([0.01, 0.03, 0.08]))
    recs = []
    for i in range(n):
        if i % fam == 0:
            L = int(np.exp(rng.uniform(np.log(150), np.log(4000))))
            tmpl = rng.choice(alpha, L)
        m = tmpl.copy()
        hits = rng.random(m.size) < rate
        m[hits] = rng.choice(alpha, int(hits.sum()))
        s = bytearray(m.tobytes())
        if rng.integers(0, 6) == 0:
            a = int(rng.integers(0, len(s)))
            del s[a:a + int(rng.integers(1, max(2, len(s) // 10)))]
        if rng.integers(0, 15) == 0:
            a = int(rng.integers(0, len(s)))
            s[a:a + 12] = b"N" * 12
        recs.append((">s%d family_%d" % (i, i // fam), bytes(s)))
    order = rng.permutation(n)
    fa = os.path.join(tmp, "in_%d.fa" % seed)
    with open(fa, "wb") as f:
        for i in order:
            h, s = recs[int(i)]
            f.write(h.encode() + b"\n")
            for a in range(0, len(s), 70):
                f.write(s[a:a + 70] + b"\n")
    outs = []
    sparse_ok = 4 ** k * dtype // 8 >= 65536
    modes = [[], ["--serial-update"]] + ([["--sparse"], ["--sparse", "--serial-update"]] if sparse_ok else [])
    for extra in modes:
        out = os.path.join(tmp, "o_%d_%d.clstr" % (seed, len(outs)))
        r = subprocess.run([EXE, fa, "--recover", os.path.join(ROOT, "tests", "golden", wts), "--id", str(ident), "--kmer", str(k), "--datatype", str(dtype),
                            "--output", out] + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        if r.returncode != 0:
            raise AssertionError("seed %d %s: msc_cluster failed\n%s" % (seed, extra, r.stdout.decode(errors="replace")[-1500:]))
        outs.append(open(out, "rb").read())
    for i in range(1, len(outs)):
        if outs[i] != outs[0]:
            raise AssertionError("seed %d: .clstr of mode %s differs from the batched dense run (%s, k=%d, u%d, n=%d)" % (seed, modes[i], wts, k, dtype, n))
    return "driver seed %d ok: %s k=%d u%d id=%.2f n=%d -> %d clusters, %d modes" % (seed, wts, k, dtype, ident, n, outs[0].count(b">Cluster"), len(outs))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t_end = time.time() + budget
    n = 0
    with tempfile.TemporaryDirectory() as tmp:
        while time.time() < t_end:
            print(run_round(seed, tmp), flush=True)
            seed += 1
            n += 1
    print("driver fuzz ok: %d rounds" % n)


if __name__ == "__main__":
    main()
