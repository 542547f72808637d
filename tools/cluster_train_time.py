#!/usr/bin/env python3
"""End-to-end timing of msc_cluster WITHOUT --recover (k rule, datatype scan, own training set, msc_train_class, clustering) -- run on
the GPU box.   python tools/cluster_train_time.py n_seqs [extra msc_cluster flags]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from meshclust2_amd import synth
n = int(sys.argv[1]); extra = sys.argv[2:]
fa = "/tmp/ct_train_%d.fa" % n
seqs, headers = synth.families(4321, n, 1000)
synth.write_fasta(fa, seqs, headers)
t0 = time.time()
out = subprocess.run([os.path.join(ROOT, "meshclust2_amd", "host", "msc_cluster"), fa, "--id", "0.9", "--output", "/tmp/ct_train.clstr"] + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
print(out.stdout.decode(errors="replace")[-2500:])
print("wall %.2f s for %d sequences" % (time.time() - t0, n))
txt = open("/tmp/ct_train.clstr").read()
print("clusters", txt.count(">Cluster"))
