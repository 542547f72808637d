#!/usr/bin/env python3
"""Randomised end-to-end rounds against the REAL reference CLI -- run where oracle/_ref/meshclust2 exists (this container, and the GPU
box: the prebuilt binary travels).   python tests/fuzz_vs_reference.py [seconds] [first seed]
Per round: a random FASTA file (families of mutated templates, mixed lengths, shuffled record order) -> the reference's own
`meshclust2` trains a model and clusters (OMP_NUM_THREADS=1; it leaves weights.txt behind) -> msc_cluster --recover weights.txt
clusters the same file on the GPU -> the two .clstr files must be the same bytes. Random k (4..9), histogram type, identity
threshold, `--feat fast|slow`; a third of the files carry N runs, IUPAC codes and lower-case records. Test infrastructure: the reference binary is the checker, exactly as for the committed fixtures."""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "meshclust2")
EXE = os.path.join(ROOT, "meshclust2_amd", "host", "msc_cluster")


def write_random_fasta(rng, path):
    n = int(rng.integers(60, 420)) * (5 if os.environ.get("FUZZ_BIG") == "1" else 1)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    fam = int(rng.integers(3, 20))
    rate = float(rng.choice([0.01, 0.03, 0.06]))
    lo, hi = (200, 900) if rng.integers(0, 2) else (400, 2500)
    dirty = rng.integers(0, 3) == 0
    repeats = not dirty and rng.integers(0, 3) == 0
    recs = []
    for i in range(n):
        if i % fam == 0:
            tmpl = rng.choice(alpha, int(np.exp(rng.uniform(np.log(lo), np.log(hi)))))
        m = tmpl.copy()
        hits = rng.random(m.size) < rate
        m[hits] = rng.choice(alpha, int(hits.sum()))
        s = bytearray(m.tobytes())
        if rng.integers(0, 8) == 0:
            a = int(rng.integers(0, len(s)))
            del s[a:a + int(rng.integers(1, max(2, len(s) // 12)))]
        if repeats and rng.integers(0, 12) == 0:    # a homopolymer / dinucleotide run: counts beyond 255, so the type scan must pick 16 bits
            a = int(rng.integers(0, len(s)))
            s[a:a] = (b"A" if rng.integers(0, 2) else b"AC") * int(rng.integers(280, 700))
        if dirty:                                   # the encoder's corner: N runs that merge / split / drop segments, IUPAC codes, lower case
            if rng.integers(0, 5) == 0:
                for _ in range(int(rng.integers(1, 4))):
                    a = int(rng.integers(0, len(s)))
                    g = int(rng.integers(1, 30))
                    s[a:a + g] = b"N" * min(g, len(s) - a)
            if rng.integers(0, 20) == 0:
                s[int(rng.integers(0, len(s)))] = int(rng.choice(np.frombuffer(b"RYMKSWHBVD", dtype=np.uint8)))
            if rng.integers(0, 10) == 0:
                s = bytearray(bytes(s).lower())
        recs.append((">r%d fam%d" % (i, i // fam), bytes(s)))
    order = [int(i) for i in rng.permutation(n)]
    if path is None:
        return [recs[i] for i in order]
    with open(path, "wb") as f:
        for i in order:
            h, s = recs[i]
            f.write(h.encode() + b"\n")
            for a in range(0, len(s), 60):
                f.write(s[a:a + 60] + b"\n")
    return n


def partition(clstr):
    """.clstr bytes -> sorted list of (sorted member headers, centre header)"""
    out, members, centre = [], [], None
    for line in clstr.decode(errors="replace").splitlines() + [">Cluster end"]:
        if line.startswith(">Cluster"):
            if members:
                out.append((tuple(sorted(members)), centre))
            members, centre = [], None
        elif line.strip():
            name = line.split(">", 1)[1].split("...")[0]
            members.append(name)
            if line.rstrip().endswith("*"):
                centre = name
    return sorted(out)


def fp_summed(weights_path):
    """does the classifier hold a statistic the reference accumulates in FP64 over the bins (jefferey 128, pearson 512, jensen_shannon 2^29)?"""
    text = open(weights_path).read().split("n_singles:")[1].splitlines()[1:]
    flags = [int(l.split()[0]) for l in text if l.strip() and l.split()[0].isdigit()]
    return any(f in (128, 512, 1 << 29) for f in flags)


def run_round(seed, tmp):
    rng = np.random.default_rng(seed)
    d = os.path.join(tmp, "r%d" % seed)
    os.makedirs(d)
    single = rng.integers(0, 8) == 0 or os.environ.get("FUZZ_SINGLE_FILE") == "1"
    if single:          # --single-file: every FASTA file is ONE sequence, its records joined by 50 N (clutil/SingleFileLoader.cpp:45-123)
        recs = write_random_fasta(rng, None)
        inputs = []
        i = 0
        while i < len(recs):
            per = int(rng.integers(1, 4))
            name = os.path.join(d, "f%03d.fa" % len(inputs))
            with open(name, "wb") as f:
                for h, sq in recs[i:i + per]:
                    f.write(h.encode() + b"\n")
                    for a in range(0, len(sq), 60):
                        f.write(sq[a:a + 60] + b"\n")
            inputs.append(name)
            i += per
        n = len(inputs)
    else:
        fa = os.path.join(d, "in.fa")
        n = write_random_fasta(rng, fa)
        inputs = [fa]
    k = int(rng.integers(4, 10))
    dtype = int(rng.choice([8, 16, 32]))
    ident = float(rng.choice([0.6, 0.8, 0.9, 0.95]))
    feat = "slow" if rng.integers(0, 3) == 0 else "fast"
    feat = os.environ.get("FUZZ_FEAT", feat)
    auto = rng.integers(0, 4) == 0 or os.environ.get("FUZZ_ALWAYS_AUTO") == "1"                  # let the reference choose k (find_k) and the histogram type itself; msc_cluster reads them from weights.txt
    id_text = str(ident)
    flags = ["--id", id_text, "--feat", feat] + ([] if auto else ["--kmer", str(k), "--datatype", str(dtype)]) + (["--single-file"] if single else [])
    if rng.integers(0, 3) == 0:                      # the mean-shift knobs: neighbourhood half-width and iteration cap
        flags += ["--delta", str(int(rng.integers(1, 9))), "--iterations", str(int(rng.integers(1, 21)))]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    t0 = time.time()
    r = subprocess.run([REF] + inputs + flags + ["--threads", "1", "--output", "ref.clstr"], cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200)
    t_ref = time.time() - t0
    if r.returncode != 0 or not os.path.exists(os.path.join(d, "weights.txt")):
        return "ref seed %d skipped: the reference exited with %d (%s)" % (seed, r.returncode, r.stdout.decode(errors="replace")[-200:].replace("\n", " | "))
    t0 = time.time()
    g = subprocess.run([EXE] + inputs + ["--recover", "weights.txt"] + flags + ["--output", "gpu.clstr"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    t_gpu = time.time() - t0
    if g.returncode != 0:
        raise AssertionError("seed %d: msc_cluster failed: %s" % (seed, g.stdout.decode(errors="replace")[-1500:]))
    a, b = open(os.path.join(d, "ref.clstr"), "rb").read(), open(os.path.join(d, "gpu.clstr"), "rb").read()
    if a != b and fp_summed(os.path.join(d, "weights.txt")) and [m for m, _ in partition(a)] == [m for m, _ in partition(b)]:
        # Same clusters; members listed in another order and, where the two members of a cluster are equally near their mean, the other one
        # starred. With a model that holds jefferey / jensen_shannon / pearson the reference adds 4^k FP64 terms per pair one by one;
        # candidates that tie mathematically (unrelated sequences of equal length share no k-mers at large k, so every statistic is a
        # function of the two lengths alone) are then ranked by the rounding noise of ITS summation order -- which of them becomes the next
        # centre, hence the visiting order -- and no other evaluation order reproduces that noise. Counted, not failed.
        pa, pb = partition(a), partition(b)
        return "ref seed %d same-clusters: k=%d u%d id=%.2f %s n=%d: %d clusters equal as sets, %d with another centre, member order differs (FP-summed statistics, tied candidates)" % (
            seed, k, dtype, ident, feat, n, len(pa), sum(1 for x, y in zip(pa, pb) if x[1] != y[1]))
    if a != b:
        keep = os.path.join(ROOT, "gpurun_out", "ref_mismatch_%d" % seed)
        os.makedirs(keep, exist_ok=True)
        for name in [os.path.basename(x) for x in inputs] + ["weights.txt", "ref.clstr", "gpu.clstr"]:
            with open(os.path.join(d, name), "rb") as src, open(os.path.join(keep, name), "wb") as dst:
                dst.write(src.read())
        raise AssertionError("seed %d: .clstr differs from the reference's (k=%d u%d id=%.2f feat=%s n=%d); files kept in %s" % (seed, k, dtype, ident, feat, n, keep))
    if auto:
        # the driver's own choice of k and histogram type (no --recover: it trains its own model, so only the two lines are compared)
        own = subprocess.run([EXE] + inputs + flags + ["--output", "own.clstr", "--dump", "own_weights.txt"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        pick = lambda text, key: [ln.strip() for ln in text.replace("\r", "\n").splitlines() if ln.startswith(key)][:1]
        ref_log, own_log = r.stdout.decode(errors="replace"), own.stdout.decode(errors="replace")
        for key in ("avg length:", "Recommended K:", "Using "):
            if pick(ref_log, key) != pick(own_log, key):
                raise AssertionError("seed %d: the driver chose differently from the reference: %r vs %r" % (seed, pick(own_log, key), pick(ref_log, key)))
        text = open(os.path.join(d, "weights.txt")).read()
        k = int(text.split("k:")[1].split()[0])
        dtype = {"uint8_t": 8, "uint16_t": 16, "uint32_t": 32, "uint64_t": 64}[text.split("Datatype:")[1].split()[0]]
    return "ref seed %d ok: k=%d u%d%s id=%s %s%s n=%d -> %d clusters (reference %.1f s, msc_cluster %.1f s)" % (seed, k, dtype, " (chosen by the reference)" if auto else "", id_text, feat, (" single-file" if single else "") + (" " + " ".join(flags[flags.index("--delta"):flags.index("--delta") + 4]) if "--delta" in flags else ""), n,
                                                                                                         a.count(b">Cluster"), t_ref, t_gpu)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    if not os.path.exists(REF):
        raise SystemExit("oracle/_ref/meshclust2 is not built (needs /root/reference at build time)")
    t_end = time.time() + budget
    n = 0
    with tempfile.TemporaryDirectory() as tmp:
        while time.time() < t_end:
            print(run_round(seed, tmp), flush=True)
            seed += 1
            n += 1
    print("reference fuzz done: %d rounds" % n)


if __name__ == "__main__":
    main()
