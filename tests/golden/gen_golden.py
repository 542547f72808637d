#!/usr/bin/env python3
"""Regenerates every fixture in tests/golden/ from the REAL reference (oracle/_ref, built by `make -C oracle ref`
from /root/reference). Run in the build container only:  python tests/golden/gen_golden.py

What is stored is DATA (inputs + the reference's outputs), never reference source:
  weights_k5_u16.txt     model trained by the reference's own `meshclust2 --dump`-style run (weights.txt) on the cfg1
                         synthetic set (classification block) + a hand-assembled regression block (the reference's
                         fastcar --dump aborts with std::bad_cast in this build, so no reference-trained one exists)
  weights_k9_u32.txt     same for k=9 / uint32_t (cfg2's model)
  cfg1.clstr             the reference CLI's CLSTR output for cfg1 (1000 x 1 kb, --id 0.9 --kmer 5 --datatype 16, 1 thread)
  vectors_*.npz          sequences, histograms (sparse: bins != 1), scalars, the 11 raw statistics for all ordered
                         pairs, model outputs, Trainer::get_close / filter / merge results, mean + distance_d
  kat_appendix_d.json    the k=2 known answers of SURVEY.md Appendix D, re-derived from the reference
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from meshclust2_amd import synth  # noqa: E402
from oracle import ref_py  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from golden_util import cfg5_set, long_fragment_set, training_set, weights_with_mode  # noqa: E402

FEATS = [("manhattan", 2), ("euclidean", 3), ("normalized_vectors", 5), ("jefferey_divergence", 7), ("pearson", 9),
         ("intersection", 13), ("emd", 18), ("length_difference", 21), ("kulczynski2", 27), ("simratio", 28), ("jensen_shannon", 29)]
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "meshclust2")

REG_BLOCK_K5 = """
n_combos: 3
0.05
0 32 0.9
1 8196 0.08
2 262148 -0.06

n_singles: 4
32 0.55 1
4 0 900
8192 0.3 1
262144 0 60000
"""

REG_BLOCK_K5_SLOW = """
n_combos: 3
0.05
0 536870912 0.9
1 8320 0.08
3 128 -0.06

n_singles: 3
536870912 0 0.02
128 0 0.2
8192 0.3 1
"""

REG_BLOCK_K9 = """
n_combos: 3
0.02
0 8192 0.95
1 268435464 0.05
3 262144 -0.04

n_singles: 4
8192 0.99 1
8 0 70
268435456 0.9 1
262144 0 40000000
"""


def run_reference_cli(fasta, args, workdir):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run([REF_BIN, fasta] + args, cwd=workdir, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, check=True)
    return out.stdout.decode(errors="replace")


def make_weights(name, seed, n, length, k, dtype, reg_block, clstr_name=None, extra_args=()):
    tmp = tempfile.mkdtemp()
    seqs, hdrs = synth.families(seed, n, length)
    fa = os.path.join(tmp, "in.fa")
    synth.write_fasta(fa, seqs, hdrs)
    run_reference_cli(fa, ["--id", "0.9", "--kmer", str(k), "--datatype", str(dtype), "--threads", "1", "--output", "out.clstr"] + list(extra_args), tmp)
    text = open(os.path.join(tmp, "weights.txt")).read()
    text = text.replace("mode: 1", "mode: 3") + reg_block
    open(os.path.join(HERE, name), "w").write(text)
    if clstr_name:
        shutil.copy(os.path.join(tmp, "out.clstr"), os.path.join(HERE, clstr_name))
    shutil.rmtree(tmp)
    print("wrote", name)


def sparse(bins):
    idx = np.nonzero(bins != 1)[0]
    return idx.astype(np.uint32), bins[idx]


def make_vectors(name, weights, seed, n, length, k, dtype, extra=()):
    seqs, _ = synth.families(seed, n, length, family=max(2, n // 3))
    seqs = list(seqs) + list(extra)
    n = len(seqs)
    pts = [ref_py.Point(dtype, s, k) for s in seqs]
    model = ref_py.Model(dtype, os.path.join(HERE, weights))
    out = {"k": k, "dtype": dtype, "n": n}
    out["seqs"] = np.array(seqs, dtype=object)
    metas = [p.meta() for p in pts]
    out["mag"] = np.array([m["mag"] for m in metas], dtype=np.uint64)
    out["length"] = np.array([m["length"] for m in metas], dtype=np.uint64)
    out["stddev"] = np.array([m["stddev"] for m in metas])
    out["one_mers"] = np.array([m["one_mers"] for m in metas], dtype=np.uint64)
    for i, p in enumerate(pts):
        ix, v = sparse(p.bins())
        out["bins_idx_%d" % i] = ix
        out["bins_val_%d" % i] = v
    raw = np.zeros((n, n, len(FEATS)))
    for i in range(n):
        for j in range(n):
            for f, (_, bit) in enumerate(FEATS):
                raw[i, j, f] = ref_py.raw_feature(1 << bit, pts[i], pts[j])      # f(first=i, second=j)
    out["raw"] = raw
    ns = len(model.score(pts[0], pts[1])[0])
    singles = np.zeros((n, n, ns))
    sums = np.zeros((n, n))
    csums = np.zeros((n, n))
    pred = np.zeros((n, n))
    close = np.zeros((n, n), dtype=np.uint8)
    for i in range(n):
        for j in range(n):
            s, _, w, cs = model.score(pts[i], pts[j])
            singles[i, j] = s
            sums[i, j] = w
            csums[i, j] = cs
            pred[i, j] = model.predict(pts[i], pts[j])
            close[i, j] = model.close(pts[i], pts[j])
    out.update(singles=singles, sums=sums, csums=csums, predict=pred, close=close)
    # Trainer operators: every point as the query against all the others, two cut-offs
    for ci, cutoff in enumerate((0.9, 0.6)):
        gc_flags = np.zeros((n, n - 1), dtype=np.uint8)
        gc_best = np.zeros((n, 3))
        flt = np.zeros((n, n - 1), dtype=np.uint8)
        mrg = np.zeros(n, dtype=np.int64)
        for q in range(n):
            cands = [pts[c] for c in range(n) if c != q]
            f, bp, bs, im = model.get_close(cutoff, pts[q], cands)
            gc_flags[q] = f
            gc_best[q] = (bp, bs, im)
            flt[q] = model.filter(cutoff, pts[q], cands)
            mrg[q] = model.merge(cutoff, pts, q, q + 1, min(n - 1, q + 6)) if q + 1 < n else 0
        out["get_close_flags_%d" % ci] = gc_flags
        out["get_close_best_%d" % ci] = gc_best
        out["filter_%d" % ci] = flt
        out["merge_%d" % ci] = mrg
    out["cutoffs"] = np.array([0.9, 0.6])
    members = list(range(0, n, 2))
    mean, d, near = ref_py.mean_nearest([pts[i] for i in members])
    out.update(mean_members=np.array(members), mean=mean, mean_dist=d, mean_nearest=near)
    # stale-mag centre: clone of 0, then set(5)
    c = pts[0].clone()
    c.set(pts[min(5, n - 1)])
    out["stale_mag"] = np.uint64(c.meta()["mag"])
    out["stale_raw"] = np.array([ref_py.raw_feature(1 << bit, c, pts[min(3, n - 1)]) for _, bit in FEATS])
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, os.path.getsize(os.path.join(HERE, name)), "bytes")


def make_kat():
    a, b = "ACGTACGTTTGACCAGTACGATCGATCGAT", "ACGTACGATTGACCAGTTCGATCGGATCGATAA"
    out = {"A": a, "B": b, "k": 2}
    for dt in (8, 16, 32, 64):
        pa, pb = ref_py.Point(dt, a, 2), ref_py.Point(dt, b, 2)
        out["u%d" % dt] = {
            "hist_A": pa.bins().tolist(), "hist_B": pb.bins().tolist(), "one_mers_A": pa.meta()["one_mers"],
            "raw": {nm: ref_py.raw_feature(1 << bit, pa, pb) for nm, bit in FEATS},
            "distance": int(ref_py.lib().ref_distance(dt, pa.h, pb.h)),
        }
    json.dump(out, open(os.path.join(HERE, "kat_appendix_d.json"), "w"), indent=1)
    print("wrote kat_appendix_d.json")


def mixed_length_set():
    """2400 sequences in three length groups (~600 / ~1000 / ~1500 bp, +-120 per family): three bvec bins, real length windows"""
    seqs, hdrs = [], []
    for gi, (n, length, seed) in enumerate(((800, 600, 31), (800, 1000, 32), (800, 1500, 33))):
        s, h = synth.families(seed, n, length, length_jitter=120)
        seqs += s
        hdrs += [">m%d_%s" % (gi, x[1:]) for x in h]
    return seqs, hdrs


def make_mixed_clstr():
    """reference CLI end to end (train + cluster, 1 thread) on the mixed-length set: its weights.txt and its .clstr"""
    tmp = tempfile.mkdtemp()
    seqs, hdrs = mixed_length_set()
    fa = os.path.join(tmp, "in.fa")
    synth.write_fasta(fa, seqs, hdrs)
    run_reference_cli(fa, ["--id", "0.8", "--kmer", "6", "--datatype", "16", "--threads", "1", "--output", "out.clstr"], tmp)
    shutil.copy(os.path.join(tmp, "weights.txt"), os.path.join(HERE, "weights_mixed_k6_u16.txt"))
    shutil.copy(os.path.join(tmp, "out.clstr"), os.path.join(HERE, "mixed.clstr"))
    shutil.rmtree(tmp)
    print("wrote mixed.clstr")


def make_mixed_slow_clstr():
    """BASELINE cfg5 in small: mixed lengths, --feat slow (all 11 statistics incl. the two divergences), --id 0.6 (min_id stays
    0.35 at exactly 0.6, cluster/CRunner.cpp:571). Reference CLI end to end, 1 thread: its weights.txt and its .clstr"""
    tmp = tempfile.mkdtemp()
    seqs, hdrs = mixed_length_set()
    seqs, hdrs = seqs[::3], hdrs[::3]
    fa = os.path.join(tmp, "in.fa")
    synth.write_fasta(fa, seqs, hdrs)
    run_reference_cli(fa, ["--id", "0.6", "--kmer", "6", "--datatype", "16", "--feat", "slow", "--threads", "1", "--output", "out.clstr"], tmp)
    shutil.copy(os.path.join(tmp, "weights.txt"), os.path.join(HERE, "weights_mixed_slow_k6_u16.txt"))
    shutil.copy(os.path.join(tmp, "out.clstr"), os.path.join(HERE, "mixed_slow.clstr"))
    shutil.rmtree(tmp)
    print("wrote mixed_slow.clstr")


def jitter_slow_set():
    """900 sequences of 1 kb +- 100 (families of 10): every accumulate step scores a real length window at k = 9 (BASELINE cfg3's shape in
    small), with a `--feat slow` model -- the window path's divergence statistics (msc_get_close_window: the rank form)"""
    return synth.families(4711, 900, 1000, family=10, length_jitter=100)


def make_jitter_slow_clstr():
    """reference CLI end to end (train + cluster, 1 thread, histogram type by its own rule): its weights.txt and its .clstr"""
    tmp = tempfile.mkdtemp()
    seqs, hdrs = jitter_slow_set()
    fa = os.path.join(tmp, "in.fa")
    synth.write_fasta(fa, seqs, hdrs)
    log = run_reference_cli(fa, ["--id", "0.8", "--kmer", "9", "--feat", "slow", "--threads", "1", "--output", "out.clstr"], tmp)
    print([ln for ln in log.splitlines() if "bit histograms" in ln or "Number of clusters" in ln])
    shutil.copy(os.path.join(tmp, "weights.txt"), os.path.join(HERE, "weights_jitter_slow_k9.txt"))
    shutil.copy(os.path.join(tmp, "out.clstr"), os.path.join(HERE, "jitter_slow.clstr"))
    shutil.rmtree(tmp)
    print("wrote jitter_slow.clstr")


def single_file_set():
    """48 FASTA files of 3 records each (members of one family); with --single-file every file is one sequence"""
    files = []
    for i in range(48):
        t = i // 4
        tmpl = synth.template(77, t, 420)
        recs = [synth.to_ascii(synth.member(77, t, 3 * (i % 4) + j, tmpl)) for j in range(3)]
        files.append(("g%02d.fa" % i, [">genome%d_contig%d template_%d" % (i, j, t) for j in range(3)], recs))
    return files


def make_single_file_clstr():
    tmp = tempfile.mkdtemp()
    names = []
    for name, hdrs, recs in single_file_set():
        synth.write_fasta(os.path.join(tmp, name), recs, hdrs)
        names.append(name)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.run([REF_BIN] + names + ["--single-file", "--id", "0.85", "--kmer", "5", "--datatype", "16", "--threads", "1", "--output", "out.clstr"],
                   cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT, check=True)
    shutil.copy(os.path.join(tmp, "weights.txt"), os.path.join(HERE, "weights_single_file_k5_u16.txt"))
    shutil.copy(os.path.join(tmp, "out.clstr"), os.path.join(HERE, "single_file.clstr"))
    shutil.rmtree(tmp)
    print("wrote single_file.clstr")


def k8_set():
    """640 sequences of ~2 kb in two length groups: k = 8, 16-bit bins (128 KiB histograms: also valid for the sparse layout)"""
    seqs, hdrs = [], []
    for gi, (n, length, seed) in enumerate(((320, 1800, 51), (320, 2400, 52))):
        s, h = synth.families(seed, n, length, family=16, length_jitter=150)
        seqs += s
        hdrs += [">e%d_%s" % (gi, x[1:]) for x in h]
    return seqs, hdrs


def make_k8_clstr():
    tmp = tempfile.mkdtemp()
    seqs, hdrs = k8_set()
    fa = os.path.join(tmp, "in.fa")
    synth.write_fasta(fa, seqs, hdrs)
    run_reference_cli(fa, ["--id", "0.85", "--kmer", "8", "--datatype", "16", "--threads", "1", "--output", "out.clstr"], tmp)
    shutil.copy(os.path.join(tmp, "weights.txt"), os.path.join(HERE, "weights_k8_u16.txt"))
    shutil.copy(os.path.join(tmp, "out.clstr"), os.path.join(HERE, "k8.clstr"))
    shutil.rmtree(tmp)
    print("wrote k8.clstr")


def k9_set():
    """BASELINE cfg3 in small: 320 sequences of 1 kb, k = 9, histogram type left to the CLI (no count exceeds 255 -> uint8_t)"""
    return synth.families(61, 320, 1000, family=16)


def make_k9_auto_clstr():
    tmp = tempfile.mkdtemp()
    seqs, hdrs = k9_set()
    fa = os.path.join(tmp, "in.fa")
    synth.write_fasta(fa, seqs, hdrs)
    log = run_reference_cli(fa, ["--id", "0.9", "--kmer", "9", "--threads", "1", "--output", "out.clstr"], tmp)
    assert "Using 8 bit histograms" in log, log[-500:]
    shutil.copy(os.path.join(tmp, "weights.txt"), os.path.join(HERE, "weights_k9_u8.txt"))
    shutil.copy(os.path.join(tmp, "out.clstr"), os.path.join(HERE, "k9_u8.clstr"))
    shutil.rmtree(tmp)
    print("wrote k9_u8.clstr")


def make_cfg5_clstr(tag="cfg5", run_cap=900):
    """BASELINE cfg5 at its stated parameters (k = 9, lengths log-uniform 500 .. 50 000, --feat slow, --id 0.6), scaled to 240
    sequences: the reference CLI end to end, 1 thread, histogram type left to its own rule -> its weights.txt and its .clstr"""
    tmp = tempfile.mkdtemp()
    seqs, hdrs = cfg5_set(run_cap=run_cap)
    fa = os.path.join(tmp, "in.fa")
    synth.write_fasta(fa, seqs, hdrs)
    log = run_reference_cli(fa, ["--id", "0.6", "--kmer", "9", "--feat", "slow", "--threads", "1", "--output", "out.clstr"], tmp)
    bits = [ln for ln in log.splitlines() if "bit histograms" in ln]
    print(tag, bits)
    shutil.copy(os.path.join(tmp, "weights.txt"), os.path.join(HERE, "weights_%s_k9.txt" % tag))
    shutil.copy(os.path.join(tmp, "out.clstr"), os.path.join(HERE, "%s.clstr" % tag))
    shutil.rmtree(tmp)
    print("wrote %s.clstr" % tag)


def make_cfg5_u16_clstr():
    """the same with tandem repeats of up to 3000 bases: counts pass 255 and the reference picks 16-bit histograms"""
    make_cfg5_clstr("cfg5_u16", 3000)


def fastcar_sets():
    db, h = synth.families(41, 300, 1000, family=10, length_jitter=150)
    q, hq = synth.families(41, 40, 1000, family=10, length_jitter=150)
    q = [x[:len(x) - 7] for x in q]
    return db, h, q, [x.replace(">seq", ">qry") for x in hq]


def make_fastcar_output():
    """reference fastcar (query x database search) with --recover on the cfg1 model, one thread -> its output file"""
    tmp = tempfile.mkdtemp()
    db, h, q, hq = fastcar_sets()
    synth.write_fasta(os.path.join(tmp, "db.fa"), db, h)
    synth.write_fasta(os.path.join(tmp, "q.fa"), q, hq)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "fastcar"), "db.fa", "--query", "q.fa", "--recover", os.path.join(HERE, "weights_k5_u16.txt"),
                    "--output", "fc_out", "--threads", "1"], cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT, check=True)
    shutil.copy(os.path.join(tmp, "fc_out0"), os.path.join(HERE, "fastcar_k5_u16.out"))
    shutil.rmtree(tmp)
    print("wrote fastcar_k5_u16.out")


def make_fastcar_mode_outputs():
    """reference fastcar with a classification-only (`mode: 1`, what meshclust2 --dump writes) and a regression-only (`mode: 2`)
    weights file: work() follows Predictor::get_mode (fastcar/FC_Runner.cpp:432,446-458)"""
    text = open(os.path.join(HERE, "weights_k5_u16.txt")).read()
    for mode in (1, 2):
        tmp = tempfile.mkdtemp()
        db, h, q, hq = fastcar_sets()
        if mode == 2:          # every pair of the length window is written: keep the file small
            db, h, q, hq = db[:60], h[:60], q[:8], hq[:8]
        synth.write_fasta(os.path.join(tmp, "db.fa"), db, h)
        synth.write_fasta(os.path.join(tmp, "q.fa"), q, hq)
        open(os.path.join(tmp, "w.txt"), "w").write(weights_with_mode(text, mode))
        env = dict(os.environ, OMP_NUM_THREADS="1")
        subprocess.run([os.path.join(ROOT, "oracle", "_ref", "fastcar"), "db.fa", "--query", "q.fa", "--recover", "w.txt", "--output", "fc_out", "--threads", "1"],
                       cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT, check=True)
        shutil.copy(os.path.join(tmp, "fc_out0"), os.path.join(HERE, "fastcar_k5_u16_mode%d.out" % mode))
        shutil.rmtree(tmp)
        print("wrote fastcar_k5_u16_mode%d.out" % mode)


def make_training(name, seed, k, dtype, feat_flags, min_feat, max_feat, ident, n_templates=40, per_template=12, length=1000):
    """BestFirstSelector::train_class of the REAL reference on labelled pairs -> tests/golden/<name> (inputs + the block it wrote)"""
    import json
    ref_py.lib().ref_set_threads(1)          # the canonical order of the open list (predict/BestFirstSelector.cpp:156-170)
    seqs, pairs = training_set(seed, n_templates, per_template, length)
    pts = [ref_py.Point(dtype, s_, k) for s_ in seqs]
    n_train = len(pairs) // 2
    text, atr, ate = ref_py.train_class(dtype, k, [pts[a] for a, b, v in pairs], [pts[b] for a, b, v in pairs], [v for a, b, v in pairs], n_train,
                                        feat_flags, min_feat, max_feat, ident)
    json.dump(dict(seed=seed, k=k, dtype=dtype, feat_flags=feat_flags, min_feat=min_feat, max_feat=max_feat, id=ident, n_templates=n_templates,
                   per_template=per_template, length=length, n_train=n_train, pairs=[[a, b, v] for a, b, v in pairs], block=text, train_acc=atr,
                   test_acc=ate), open(os.path.join(HERE, name), "w"))
    print("wrote", name, "train/test accuracy", atr, ate)


def make_vectors_k13():
    """BASELINE cfg4 at its stated parameters, printed by the reference itself: k = 13, uint64_t histograms (512 MiB each, 1.5 GiB while
    Loader::get_point copies them: clutil/Loader.cpp:138-179), 20 kb sequences -- the five of tests' _cfg4_sequences() plus the relative
    with a 70 000-base homopolymer run (one bin of 69 988 >= 2^16). Stored: the sequences, the bins that differ from the pseudocount,
    mag / length / stddev / 1-mers, the 11 raw statistics of every ordered pair (predict/Feature.cpp:682-1518), Trainer::get_close and
    filter under weights_cfg4_k13.txt for two cut-offs, mean + distance_d of three members. The model file is hand-written (the
    reference cannot train at this size here: ~4 800 mutant histograms of 512 MiB); every number below it is the reference's."""
    from golden_util import cfg4_sequences
    k, dtype = 13, 64
    seqs, mono = cfg4_sequences()
    seqs = list(seqs) + [mono]
    n = len(seqs)
    ref_py.lib().ref_set_threads(8)
    pts = [ref_py.Point(dtype, s, k) for s in seqs]
    out = {"k": k, "dtype": dtype, "n": n, "seqs": np.array(seqs, dtype=object)}
    metas = [p.meta() for p in pts]
    out["mag"] = np.array([m["mag"] for m in metas], dtype=np.uint64)
    out["length"] = np.array([m["length"] for m in metas], dtype=np.uint64)
    out["stddev"] = np.array([m["stddev"] for m in metas])
    out["one_mers"] = np.array([m["one_mers"] for m in metas], dtype=np.uint64)
    for i, p in enumerate(pts):
        ix, v = sparse(p.bins())
        out["bins_idx_%d" % i] = ix
        out["bins_val_%d" % i] = v
    raw = np.zeros((n, n, len(FEATS)))
    for i in range(n):
        for j in range(n):
            for f, (_, bit) in enumerate(FEATS):
                raw[i, j, f] = ref_py.raw_feature(1 << bit, pts[i], pts[j])
        print("k13 raw statistics of point", i, flush=True)
    out["raw"] = raw
    model = ref_py.Model(dtype, os.path.join(HERE, "weights_cfg4_k13.txt"))
    ref_py.lib().ref_set_threads(1)          # the canonical arg-max order of Trainer::get_close (cluster/Trainer.cpp:41)
    for ci, cutoff in enumerate((0.9, 0.6)):
        gc_flags = np.zeros((n, n - 1), dtype=np.uint8)
        gc_best = np.zeros((n, 3))
        flt = np.zeros((n, n - 1), dtype=np.uint8)
        for q in range(n):
            cands = [pts[c] for c in range(n) if c != q]
            f, bp, bs, im = model.get_close(cutoff, pts[q], cands)
            gc_flags[q] = f
            gc_best[q] = (bp, bs, im)
            flt[q] = model.filter(cutoff, pts[q], cands)
        out["get_close_flags_%d" % ci] = gc_flags
        out["get_close_best_%d" % ci] = gc_best
        out["filter_%d" % ci] = flt
        print("k13 operators at cut-off", cutoff, flush=True)
    out["cutoffs"] = np.array([0.9, 0.6])
    sums = np.zeros((n, n))
    for i in range(n):
        for j in range(n):
            sums[i, j] = model.score(pts[i], pts[j])[2]
    out["sums"] = sums
    members = [0, 1, 2]
    mean, d, near = ref_py.mean_nearest([pts[i] for i in members])
    ix = np.nonzero(mean != 1.0)[0]
    out.update(mean_members=np.array(members), mean_idx=ix.astype(np.uint32), mean_val=mean[ix], mean_dist=d, mean_nearest=near)
    np.savez_compressed(os.path.join(HERE, "vectors_k13_u64.npz"), **out)
    print("wrote vectors_k13_u64.npz", os.path.getsize(os.path.join(HERE, "vectors_k13_u64.npz")), "bytes")


def make_training_regr(name, seed, k, dtype, feat_flags, max_feat, ident, n_templates=40, per_template=12, length=1000):
    """Predictor<T>::train_regr's selection (GreedySelector::train_regression, followed on the REAL reference's objects: oracle/ref_harness.cpp
    train_regr -- the function itself has no return statement and cannot be run to its end) on labelled pairs with identity above `ident`
    (Predictor::train keeps those for the regression, predict/Predictor.cpp:923-927) -> tests/golden/<name> (inputs + the block it printed)"""
    ref_py.lib().ref_set_threads(1)
    seqs, pairs = training_set(seed, n_templates, per_template, length)
    pairs = [p for p in pairs if p[2] > ident]
    pts = [ref_py.Point(dtype, s_, k) for s_ in seqs]
    n_train = len(pairs) // 2
    text, etr, ete = ref_py.train_regr(dtype, k, [pts[a] for a, b, v in pairs], [pts[b] for a, b, v in pairs], [v for a, b, v in pairs], n_train, feat_flags, max_feat)
    json.dump(dict(seed=seed, k=k, dtype=dtype, feat_flags=feat_flags, max_feat=max_feat, id=ident, n_templates=n_templates, per_template=per_template,
                   length=length, n_train=n_train, pairs=[[a, b, v] for a, b, v in pairs], block=text, train_err=etr, test_err=ete),
              open(os.path.join(HERE, name), "w"))
    print("wrote", name, "train/test mean error", etr, ete)
    return text


def make_regr_weights_and_fastcar():
    """A `mode: 3` weights file whose BOTH blocks were printed by the reference: the classification block of weights_k5_u16.txt (trained by
    the reference CLI) + the regression block of train_regr_k5_u16.json (above); and the reference fastcar's output with it on the same
    database / query files as fastcar_k5_u16.out."""
    block = json.load(open(os.path.join(HERE, "train_regr_k5_u16.json")))["block"]
    cls = weights_with_mode(open(os.path.join(HERE, "weights_k5_u16.txt")).read(), 1)
    text = cls.replace("mode: 1", "mode: 3").rstrip("\n") + "\n" + block
    open(os.path.join(HERE, "weights_k5_u16_regr.txt"), "w").write(text)
    tmp = tempfile.mkdtemp()
    db, h, q, hq = fastcar_sets()
    synth.write_fasta(os.path.join(tmp, "db.fa"), db, h)
    synth.write_fasta(os.path.join(tmp, "q.fa"), q, hq)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "fastcar"), "db.fa", "--query", "q.fa", "--recover", os.path.join(HERE, "weights_k5_u16_regr.txt"),
                    "--output", "fc_out", "--threads", "1"], cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT, check=True)
    shutil.copy(os.path.join(tmp, "fc_out0"), os.path.join(HERE, "fastcar_k5_u16_regr.out"))
    shutil.rmtree(tmp)
    print("wrote weights_k5_u16_regr.txt and fastcar_k5_u16_regr.out")


def fastcar_k9_sets():
    """cfg2's shape for the search: 1 kb relatives at k = 9; every ninth database sequence and one query carry a run (a 300-base
    homopolymer, 150 x `AC`, or a 12-mer x 10) -- counts of 11 .. 290 among counts of 1 .. 3, as real FASTA has them"""
    db, h = synth.families(43, 220, 1000, family=10, length_jitter=120)
    q, hq = synth.families(43, 30, 1000, family=10, length_jitter=120)
    runs = [b"A" * 300, b"AC" * 150, b"ACGTTGCAAGTC" * 10]
    db = [s[:200 + i] + runs[(i // 9) % 3] + s[200 + i:] if i % 9 == 4 else s for i, s in enumerate(db)]
    q = [x[:len(x) - 5] for x in q]
    q[4] = q[4][:333] + runs[0] + q[4][333:]
    return db, h, q, [x.replace(">seq", ">qry") for x in hq]


REG_BLOCK_K9_FC = """
n_combos: 2
0.03
0 8192 0.9
3 262144 0.06

n_singles: 2
8192 0 1
262144 0 90000000
"""


def make_fastcar_k9_output():
    """reference fastcar at k = 9 / uint32_t (the histogram shape of BASELINE cfg2) on a database with repeat-bearing sequences: the
    classification block the reference trained (weights_k9_u32.txt) + a regression block over intersection and emd whose values spread
    over (0, 1), so the third column of the output is not all 100"""
    cls = weights_with_mode(open(os.path.join(HERE, "weights_k9_u32.txt")).read(), 1)
    text = cls.replace("mode: 1", "mode: 3").rstrip("\n") + "\n" + REG_BLOCK_K9_FC
    open(os.path.join(HERE, "weights_k9_u32_fc.txt"), "w").write(text)
    tmp = tempfile.mkdtemp()
    db, h, q, hq = fastcar_k9_sets()
    synth.write_fasta(os.path.join(tmp, "db.fa"), db, h)
    synth.write_fasta(os.path.join(tmp, "q.fa"), q, hq)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "fastcar"), "db.fa", "--query", "q.fa", "--recover", os.path.join(HERE, "weights_k9_u32_fc.txt"),
                    "--output", "fc_out", "--threads", "1"], cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT, check=True)
    shutil.copy(os.path.join(tmp, "fc_out0"), os.path.join(HERE, "fastcar_k9_u32.out"))
    shutil.rmtree(tmp)
    print("wrote weights_k9_u32_fc.txt and fastcar_k9_u32.out")


def make_long_fragments():
    seqs = long_fragment_set()
    out = {"n": len(seqs), "lengths": np.array([len(s) for s in seqs])}
    for i, s in enumerate(seqs):
        codes, segs, eff = ref_py.encode(s)
        out["segs_%d" % i] = np.array(segs, dtype=np.int64)
        out["eff_%d" % i] = eff
        for k, dt in ((5, 32), (7, 16)):
            p = ref_py.Point(dt, s, k)
            m = p.meta()
            out["bins_k%d_%d" % (k, i)] = p.bins()
            out["meta_k%d_%d" % (k, i)] = np.array([m["mag"], m["length"]] + m["one_mers"], dtype=np.uint64)
            out["stddev_k%d_%d" % (k, i)] = m["stddev"]
    np.savez_compressed(os.path.join(HERE, "long_fragments.npz"), **out)
    print("wrote long_fragments.npz", [(len(s), out["segs_%d" % i].tolist()) for i, s in enumerate(seqs)])


FAST_FLAGS = sum(1 << b for b in (2, 3, 5, 9, 13, 18, 21, 27, 28))
SLOW_FLAGS = FAST_FLAGS | (1 << 7) | (1 << 29)

NASTY = [
    b"ACGTNNNNACGTACGTACGTACGTAACCGGTTNNNNNNNNNNNNACGATCGATCGATCGATCGACTAGCTAGCTAGCATCGAT" * 6,
    b"acgtacgtnnacgtRYMKSWHBVDacgtacgtacgtagctagcatcgatcgatcgatcagctagcat" * 9,
]

if __name__ == "__main__":
    if not ref_py.available():
        sys.exit("oracle/_ref is not built: run `make -C oracle ref` (needs /root/reference)")
    if len(sys.argv) > 1:          # regenerate single fixtures: python gen_golden.py make_cfg5_clstr ...
        for fn in sys.argv[1:]:
            globals()[fn]()
        sys.exit(0)
    make_kat()
    make_training("train_k5_u16.json", 31, 5, 16, FAST_FLAGS, 4, 4, 0.9)
    make_training("train_k7_u8_slow.json", 32, 7, 8, SLOW_FLAGS, 2, 3, 0.8, n_templates=30, per_template=10, length=600)
    make_training("train_k9_u32.json", 33, 9, 32, FAST_FLAGS, 3, 4, 0.9, n_templates=24, per_template=10)
    make_training_regr("train_regr_k5_u16.json", 41, 5, 16, FAST_FLAGS, 4, 0.7)
    make_training_regr("train_regr_k7_u8_slow.json", 42, 7, 8, SLOW_FLAGS, 3, 0.75, n_templates=30, per_template=10, length=600)
    make_mixed_clstr()
    make_mixed_slow_clstr()
    make_k9_auto_clstr()
    make_weights("weights_k5_u16.txt", 20260001, 1000, 1000, 5, 16, REG_BLOCK_K5, clstr_name="cfg1.clstr")
    make_weights("weights_k9_u32.txt", 20260002, 300, 1000, 9, 32, REG_BLOCK_K9)
    make_weights("weights_k5_u16_slow.txt", 20260001, 1000, 1000, 5, 16, REG_BLOCK_K5_SLOW, extra_args=["--feat", "slow"])
    make_vectors("vectors_k5_u16_slow.npz", "weights_k5_u16_slow.txt", 15, 12, 1000, 5, 16, extra=NASTY)
    make_fastcar_output()
    make_fastcar_mode_outputs()
    make_regr_weights_and_fastcar()
    make_fastcar_k9_output()
    make_long_fragments()
    make_jitter_slow_clstr()
    make_cfg5_clstr()
    make_cfg5_u16_clstr()
    make_k8_clstr()
    make_single_file_clstr()
    make_vectors("vectors_k5_u16.npz", "weights_k5_u16.txt", 11, 18, 1000, 5, 16, extra=NASTY)
    make_vectors("vectors_k9_u32.npz", "weights_k9_u32.txt", 12, 8, 1000, 9, 32)
    make_vectors("vectors_k4_u8.npz", "weights_k5_u16.txt", 13, 10, 150, 4, 8)
    make_vectors("vectors_k6_u64.npz", "weights_k5_u16.txt", 14, 8, 1500, 6, 64)
    make_vectors_k13()
