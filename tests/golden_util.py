"""Shared helpers for the golden fixtures in tests/golden/ (generated from the real reference by gen_golden.py)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FEATS = [("manhattan", 2), ("euclidean", 3), ("normalized_vectors", 5), ("jefferey_divergence", 7), ("pearson", 9),
         ("intersection", 13), ("emd", 18), ("length_difference", 21), ("kulczynski2", 27), ("simratio", 28), ("jensen_shannon", 29)]
FEAT_BIT = dict(FEATS)
# statistics whose value is an exact integer reduction pushed through the same FP64 expression: compared bitwise
EXACT = {"manhattan", "euclidean", "normalized_vectors", "intersection", "emd", "length_difference", "kulczynski2", "simratio"}
FAST = [f for f in FEATS if f[0] not in ("jefferey_divergence", "jensen_shannon")]
VECTOR_SETS = [("vectors_k5_u16.npz", "weights_k5_u16.txt"), ("vectors_k5_u16_slow.npz", "weights_k5_u16_slow.txt"), ("vectors_k9_u32.npz", "weights_k9_u32.txt"),
               ("vectors_k4_u8.npz", "weights_k5_u16.txt"), ("vectors_k6_u64.npz", "weights_k5_u16.txt")]
NP_T = {8: np.uint8, 16: np.uint16, 32: np.uint32, 64: np.uint64}


def load_vectors(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=True)
    return {k: z[k] for k in z.files}


def dense_bins(v, i):
    bins = np.ones(4 ** int(v["k"]), dtype=NP_T[int(v["dtype"])])
    bins[v["bins_idx_%d" % i]] = v["bins_val_%d" % i]
    return bins


def weights_text(name):
    return open(os.path.join(GOLDEN, name)).read()


def kat():
    return json.load(open(os.path.join(GOLDEN, "kat_appendix_d.json")))


def single_file_set():
    """48 FASTA files of 3 records each (same generator as tests/golden/gen_golden.py)"""
    from meshclust2_amd import synth
    files = []
    for i in range(48):
        t = i // 4
        tmpl = synth.template(77, t, 420)
        recs = [synth.to_ascii(synth.member(77, t, 3 * (i % 4) + j, tmpl)) for j in range(3)]
        files.append(("g%02d.fa" % i, [">genome%d_contig%d template_%d" % (i, j, t) for j in range(3)], recs))
    return files


def k8_set():
    """640 sequences of ~2 kb in two length groups (same generator as tests/golden/gen_golden.py)"""
    from meshclust2_amd import synth
    seqs, hdrs = [], []
    for gi, (n, length, seed) in enumerate(((320, 1800, 51), (320, 2400, 52))):
        s, h = synth.families(seed, n, length, family=16, length_jitter=150)
        seqs += s
        hdrs += [">e%d_%s" % (gi, x[1:]) for x in h]
    return seqs, hdrs
