#!/usr/bin/env python3
"""Train a model for msc_cluster / msc_fastcar on the GPU box, without the reference:

  python tools/train_model.py input.fa --id 0.9 --kmer 9 --datatype 32 [--feat fast|slow] [--templates 300] [--mutants 8] -o weights.txt

Templates are sampled from the input; each gets mutants at graded divergence around --id (substitutions and single-base
indels from the repo's own SplitMix64 generator, labelled with the intended identity); msc_train_class picks the combos and
fits the GLM exactly as the reference's BestFirstSelector does on ITS pairs. The pair generator is NOT the reference's
(predict/Predictor.cpp:519-710: LCG + mt19937 templates/mutation), so the model differs from what `meshclust2 --dump` would
write for the same file; everything downstream of the pairs is the reference's procedure (tests: test_training_selects_the_reference_model).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from meshclust2_amd import api, synth  # noqa: E402

FAST = sum(1 << b for b in (2, 3, 5, 9, 13, 18, 21, 27, 28))
SLOW = FAST | (1 << 7) | (1 << 29)
CODE = np.full(256, 255, dtype=np.uint8)
for i, c in enumerate(b"ACGT"):
    CODE[c] = i
    CODE[ord(chr(c).lower())] = i


def read_fasta(path):
    seqs, cur = [], []
    for line in open(path, "rb"):
        line = line.strip()
        if line.startswith(b">"):
            if cur:
                seqs.append(b"".join(cur))
            cur = []
        elif line:
            cur.append(line)
    if cur:
        seqs.append(b"".join(cur))
    return seqs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fasta")
    ap.add_argument("--id", type=float, default=0.9)
    ap.add_argument("--kmer", "-k", type=int, required=True)
    ap.add_argument("--datatype", type=int, default=32)
    ap.add_argument("--feat", choices=("fast", "slow"), default="fast")
    ap.add_argument("--templates", type=int, default=300)
    ap.add_argument("--mutants", type=int, default=8, help="mutants per template (half above, half below --id)")
    ap.add_argument("--min-feat", type=int, default=4)
    ap.add_argument("--max-feat", type=int, default=4)
    ap.add_argument("--seed", type=int, default=0xAA)
    ap.add_argument("--output", "-o", default="weights.txt")
    args = ap.parse_args()
    seqs = [s for s in read_fasta(args.fasta) if len(s) >= 50]
    rng = np.random.default_rng(args.seed)
    pick = rng.permutation(len(seqs))[: args.templates]
    ident = args.id if args.id <= 1 else args.id / 100.0
    lo = max(0.35, ident - 0.25)              # the reference's min_id idea (cluster/CRunner.cpp:571-573): negatives well below the cut-off
    points, pairs = [], []
    for t, si in enumerate(pick):
        codes = CODE[np.frombuffer(seqs[si].upper(), dtype=np.uint8)]
        codes = codes[codes < 4]
        points.append(synth.to_ascii(codes))
        ti = len(points) - 1
        for j in range(args.mutants):
            target = rng.uniform(ident, 1.0) if j % 2 == 0 else rng.uniform(lo, ident)
            rate = 1.0 - target
            points.append(synth.to_ascii(synth.member(args.seed, t, j, codes, sub_rate=rate * 0.8, indel_rate=rate * 0.2)))
            pairs.append((ti, len(points) - 1, target))
    pairs = [pairs[i] for i in rng.permutation(len(pairs))]
    n_train = len(pairs) // 2
    ctx = api.Context(0)
    pts = api.HistogramSet(ctx, args.kmer, args.datatype, len(points))
    pts.build(points)
    text, atr, ate = api.train_class(ctx, pts, [p[0] for p in pairs], [p[1] for p in pairs], [p[2] for p in pairs], n_train,
                                     FAST if args.feat == "fast" else SLOW, args.min_feat, args.max_feat, ident)
    open(args.output, "w").write(text)
    print("trained on %d + %d pairs from %d templates: training accuracy %.2f %%, testing accuracy %.2f %% -> %s" % (n_train, len(pairs) - n_train, len(pick), atr, ate, args.output))


if __name__ == "__main__":
    main()
