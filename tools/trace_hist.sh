#!/bin/bash
# Which runtime calls a driver run issues, by source line: MSC_TRACE_CALLS output of msc_cluster folded into a histogram.
#   tools/trace_hist.sh <n_seqs> <k> <dtype> <weights> [extra msc_cluster flags]      (CLUSTER_TIME_JITTER=j: lengths 1000 +- j)
N=$1; K=$2; DT=$3; W=$4; shift 4
R=${GRAFT_REPO_ROOT:-/root/repo}
python3 - <<PY
import sys, os
sys.path.insert(0, "$R")
from meshclust2_amd import synth
j = int(os.environ.get("CLUSTER_TIME_JITTER", "0"))
seqs, headers = synth.families(777, $N, 1000, length_jitter=j) if j else synth.families(777, $N, 1000)
synth.write_fasta("/tmp/th_$N.fa", seqs, headers)
PY
MSC_TRACE_CALLS=1 $R/meshclust2_amd/host/msc_cluster /tmp/th_$N.fa --recover $R/$W --id 0.9 --kmer $K --datatype $DT --output /tmp/th.clstr "$@" 2>&1 >/dev/null | grep "^\[msc\]" | awk '{print $2, $3}' | cut -c1-90 | sort | uniq -c | sort -rn | head -40
