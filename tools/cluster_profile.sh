#!/bin/bash
# Where the mean-shift driver's time goes on the GPU box: writes the synthetic FASTA, runs msc_cluster plain (its own
# "timestamp" lines) and under rocprofv3 --kernel-trace --stats (calls and mean duration per kernel).
#   tools/cluster_profile.sh <tag> <n_seqs> <k> <dtype> <weights> [extra msc_cluster flags]      (CLUSTER_TIME_JITTER=j: lengths 1000 +- j)
set -e
TAG=$1; N=$2; K=$3; DT=$4; W=$5; shift 5
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 - <<PY
import sys
sys.path.insert(0, "$R")
from meshclust2_amd import synth
import os
j = int(os.environ.get("CLUSTER_TIME_JITTER", "0"))
seqs, headers = synth.families(777, $N, 1000, length_jitter=j) if j else synth.families(777, $N, 1000)
synth.write_fasta("/tmp/cp_$N.fa", seqs, headers)
PY
$R/meshclust2_amd/host/msc_cluster /tmp/cp_$N.fa --recover $R/$W --id 0.9 --kmer $K --datatype $DT --output /tmp/cp.clstr "$@" > $O/plain.log 2>&1
grep -E "timestamp|Number of clusters" $O/plain.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- $R/meshclust2_amd/host/msc_cluster /tmp/cp_$N.fa --recover $R/$W --id 0.9 --kmer $K --datatype $DT --output /tmp/cp2.clstr "$@" > $O/stats.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open("$O/kernel_stats.csv", "w") as out:
    out.write("kernel,calls,total_ns,average_ns,percentage\n")
    for r in rows:
        name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        out.write('"%s",%s,%s,%s,%s\n' % (name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
print(open("$O/kernel_stats.csv").read()[:3000])
PY
find $O/stats -type f ! -name "*kernel_stats.csv" -delete
