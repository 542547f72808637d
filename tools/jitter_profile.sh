#!/bin/bash
# The accumulate loop on a cfg3-shaped set WITH length windows (1 kb +- 100): the driver's and the library's timers, then the kernels of
# the same run under rocprofv3 --kernel-trace --stats.        tools/jitter_profile.sh <tag> <n_seqs> [msc_cluster flags ...]   -> gpurun_out/<tag>/
set -e
TAG=$1; N=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 - <<PY
import sys
sys.path.insert(0, "$R")
from meshclust2_amd import synth
seqs, hdrs = synth.families(777, $N, 1000, length_jitter=100)
synth.write_fasta("/tmp/jit_$N.fa", seqs, hdrs)
PY
W=${JITTER_WEIGHTS:-$R/tests/golden/weights_k9_u8.txt}
ARGS="/tmp/jit_$N.fa --recover $W --id 0.9 --kmer 9 --datatype ${JITTER_DTYPE:-8}"
MSC_CLUSTER_PROFILE=1 MSC_PROFILE_CALLS=1 $R/meshclust2_amd/host/msc_cluster $ARGS --output /tmp/jit.clstr "$@" > $O/plain.log 2>&1
grep -E "timestamp|Number of clusters|profile|\[msc\]" $O/plain.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- $R/meshclust2_amd/host/msc_cluster $ARGS --output /tmp/jit2.clstr "$@" > $O/stats.log 2>&1
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
find $O/stats -type f -delete
