#!/usr/bin/env python3
"""BASELINE cfg4's shape end to end on one GPU: 20 kb sequences, k = 13, 64-bit counts, sparse layout (a dense slot would be 512 MiB);
msc_cluster trains its own model first (no fixture model exists at k = 13) -- run on the GPU box.
   python tools/cfg4_time.py [n_seqs] [msc_cluster flags ...]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from meshclust2_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
extra = sys.argv[2:]
t0 = time.time()
seqs, hdrs = synth.families(1313, n, 20000, family=10, length_jitter=2000)
fa = "/tmp/cfg4_%d.fa" % n
synth.write_fasta(fa, seqs, hdrs)
print("generated %d sequences, %.1f Mb in %.1f s" % (n, sum(len(s) for s in seqs) / 1e6, time.time() - t0), flush=True)
t0 = time.time()
out = subprocess.run([os.path.join(ROOT, "meshclust2_amd", "host", "msc_cluster"), fa, "--id", "0.9", "--kmer", "13", "--datatype", "64", "--sparse",
                      "--output", "/tmp/cfg4_time.clstr"] + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1000)
print(out.stdout.decode(errors="replace")[-1500:])
print("wall %.2f s for %d sequences" % (time.time() - t0, n))
