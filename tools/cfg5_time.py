#!/usr/bin/env python3
"""BASELINE cfg5's shape end to end on one GPU: mixed lengths 500 b - 50 kb, k = 9, 16-bit histograms, a `--feat slow` model
(tests/golden/weights_cfg5_u16_k9.txt, trained by the reference itself on the small cfg5 fixture), --id 0.6 -- run on the GPU box.
   python tools/cfg5_time.py [n_templates] [per_template] [msc_cluster flags ...]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import cfg5_set
from meshclust2_amd import synth
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 400
per = int(sys.argv[2]) if len(sys.argv) > 2 else 5
extra = sys.argv[3:]
t0 = time.time()
seqs, hdrs = cfg5_set(n_templates=nt, per_template=per, run_cap=3000)
fa = "/tmp/cfg5_%d.fa" % (nt * per)
synth.write_fasta(fa, seqs, hdrs)
bases = sum(len(s) for s in seqs)
print("generated %d sequences, %.1f Mb in %.1f s" % (len(seqs), bases / 1e6, time.time() - t0), flush=True)
prof = []
if os.environ.get("CFG5_PROFILE"):      # per-kernel calls and time of the same run (rocprofv3 --kernel-trace --stats, csv under /tmp/cfg5_prof)
    prof = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", "/tmp/cfg5_prof", "-o", "stats", "--"]
t0 = time.time()
out = subprocess.run(prof + [os.path.join(ROOT, "meshclust2_amd", "host", "msc_cluster"), fa, "--recover", os.path.join(ROOT, "tests", "golden", "weights_cfg5_u16_k9.txt"),
                      "--id", "0.6", "--output", "/tmp/cfg5_time.clstr"] + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1100,
                     cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
print(out.stdout.decode(errors="replace")[-1200:])
print("wall %.2f s for %d sequences (%s)" % (time.time() - t0, len(seqs), " ".join(extra) or "dense"))
if prof:
    import csv, glob
    f = glob.glob("/tmp/cfg5_prof/**/*kernel_stats.csv", recursive=True)[0]
    for r in list(csv.DictReader(open(f)))[:14]:
        print("%-60s calls %8s  total %10.1f ms  avg %9.1f us" % (r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60], r["Calls"],
                                                                  float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
