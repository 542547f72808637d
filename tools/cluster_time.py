#!/usr/bin/env python3
"""End-to-end timing of the mean-shift driver (meshclust2_amd/host/msc_cluster) on a synthetic set -- run on the GPU box.
   python tools/cluster_time.py [n_seqs] [k] [dtype] [weights file] [msc_cluster flags ...]      (CLUSTER_TIME_JITTER=j: lengths 1000 +- j)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from meshclust2_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dtype = sys.argv[3] if len(sys.argv) > 3 else "16"
wts = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] else os.path.join(ROOT, "tests", "golden", "weights_k8_u16.txt")
extra = sys.argv[5:]
fa = "/tmp/cluster_time_%d.fa" % n
jitter = int(os.environ.get("CLUSTER_TIME_JITTER", "0"))
t0 = time.time()
seqs, headers = synth.families(777, n, 1000, length_jitter=jitter) if jitter else synth.families(777, n, 1000)
synth.write_fasta(fa, seqs, headers)
print("generated %d sequences in %.1f s" % (n, time.time() - t0), flush=True)
t0 = time.time()
out = subprocess.run([os.path.join(ROOT, "meshclust2_amd", "host", "msc_cluster"), fa, "--recover", wts, "--id", "0.9", "--kmer", str(k), "--datatype", dtype,
                      "--output", "/tmp/cluster_time.clstr"] + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1100)
print(out.stdout.decode(errors="replace")[-1500:])
print("wall %.2f s for %d sequences (k=%d, u%s)" % (time.time() - t0, n, k, dtype))
