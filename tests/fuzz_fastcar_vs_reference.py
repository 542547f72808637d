#!/usr/bin/env python3
"""Randomised search rounds against the REAL reference `fastcar` -- run where oracle/_ref/fastcar exists (this container and the GPU
box).   python tests/fuzz_fastcar_vs_reference.py [seconds] [first seed]
Per round: a random database and query FASTA pair -> the reference's `fastcar --recover <fixture model>` on one thread ->
msc_fastcar with the same model on the GPU (random --query-block) -> the output files must be the same bytes. The models are
the committed two-block fixtures (classifier trained by the reference, regression block hand-assembled: its own fastcar --dump
aborts in this build)."""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "fastcar")
EXE = os.path.join(ROOT, "meshclust2_amd", "host", "msc_fastcar")
GOLDEN = os.path.join(ROOT, "tests", "golden")
MODELS = [f for f in sorted(os.listdir(GOLDEN)) if f.startswith("weights_") and open(os.path.join(GOLDEN, f)).read().count("n_combos") == 2]


def write_fasta(path, recs):
    with open(path, "wb") as f:
        for h, s in recs:
            f.write(h.encode() + b"\n")
            for a in range(0, len(s), 60):
                f.write(s[a:a + 60] + b"\n")


def run_round(seed, tmp):
    rng = np.random.default_rng(seed)
    d = os.path.join(tmp, "r%d" % seed)
    os.makedirs(d)
    model = MODELS[int(rng.integers(0, len(MODELS)))]
    big = "k9" in model
    n_db, n_q = int(rng.integers(30, 120 if big else 400)), int(rng.integers(3, 12 if big else 50))
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    fam = int(rng.integers(3, 15))
    rate = float(rng.choice([0.01, 0.04, 0.08]))
    db = []
    tmpls = []
    for i in range(n_db):
        if i % fam == 0:
            tmpls.append(rng.choice(alpha, int(np.exp(rng.uniform(np.log(200), np.log(1500))))))
        m = tmpls[-1].copy()
        hits = rng.random(m.size) < rate
        m[hits] = rng.choice(alpha, int(hits.sum()))
        db.append((">d%d fam%d" % (i, i // fam), m.tobytes()))
    qs = []
    for i in range(n_q):
        m = tmpls[int(rng.integers(0, len(tmpls)))].copy() if rng.integers(0, 4) else rng.choice(alpha, int(rng.integers(200, 1500)))
        hits = rng.random(m.size) < rate
        m[hits] = rng.choice(alpha, int(hits.sum()))
        s = m.tobytes()
        qs.append((">q%d" % i, s[: len(s) - int(rng.integers(0, 9))]))
    write_fasta(os.path.join(d, "db.fa"), [db[int(i)] for i in rng.permutation(n_db)])
    write_fasta(os.path.join(d, "q.fa"), qs)
    wpath = os.path.join(GOLDEN, model)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    t0 = time.time()
    r = subprocess.run([REF, "db.fa", "--query", "q.fa", "--recover", wpath, "--output", "ref_", "--threads", "1"], cwd=d, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=1200)
    t_ref = time.time() - t0
    if r.returncode != 0 or not os.path.exists(os.path.join(d, "ref_0")):
        return "fastcar seed %d skipped: the reference exited with %d (%s)" % (seed, r.returncode, r.stdout.decode(errors="replace")[-160:].replace("\n", " | "))
    qb = int(rng.choice([1, 3, 16, 32]))
    extra = ["--sparse"] if big and rng.integers(0, 2) else []          # the sparse layout exists from 64 KiB histograms up
    g = subprocess.run([EXE, "db.fa", "--query", "q.fa", "--recover", wpath, "--output", "gpu_", "--query-block", str(qb)] + extra, cwd=d, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    if g.returncode != 0:
        raise AssertionError("seed %d: msc_fastcar failed: %s" % (seed, g.stdout.decode(errors="replace")[-1500:]))
    a, b = open(os.path.join(d, "ref_0"), "rb").read(), open(os.path.join(d, "gpu_0"), "rb").read()
    if a != b:
        keep = os.path.join(ROOT, "gpurun_out", "fastcar_mismatch_%d" % seed)
        os.makedirs(keep, exist_ok=True)
        for name in ("db.fa", "q.fa", "ref_0", "gpu_0"):
            with open(os.path.join(d, name), "rb") as src, open(os.path.join(keep, name), "wb") as dst:
                dst.write(src.read())
        raise AssertionError("seed %d: search output differs from the reference's (%s, %d x %d, block %d); files kept in %s" % (seed, model, n_q, n_db, qb, keep))
    return "fastcar seed %d ok: %s %d queries x %d, block %d%s -> %d lines (reference %.1f s)" % (seed, model, n_q, n_db, qb, " sparse" if extra else "", a.count(b"\n"), t_ref)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    if not os.path.exists(REF):
        raise SystemExit("oracle/_ref/fastcar is not built (needs /root/reference at build time)")
    t_end = time.time() + budget
    n = 0
    with tempfile.TemporaryDirectory() as tmp:
        while time.time() < t_end:
            print(run_round(seed, tmp), flush=True)
            seed += 1
            n += 1
    print("fastcar fuzz ok: %d rounds" % n)


if __name__ == "__main__":
    main()
