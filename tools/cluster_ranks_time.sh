#!/bin/bash
# Wall time of the sharded driver (msc_cluster, one process per rank: host/msc_sharded.hpp) next to the one-GPU run of the same binary
# on the same file -- run on the GPU box. The ranks share device 0 and exchange over sockets + host staging (MSC_COMM=tcp): the
# exchange logic and its host cost, not xGMI.
#   tools/cluster_ranks_time.sh <n_seqs> <k> <dtype> <weights> <ranks> [msc_cluster flags ...]      (CLUSTER_TIME_JITTER=j: lengths 1000 +- j)
N=$1; K=$2; DT=$3; W=$4; RANKS=$5; shift 5
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 - <<PY
import sys, os
sys.path.insert(0, "$R")
from meshclust2_amd import synth
j = int(os.environ.get("CLUSTER_TIME_JITTER", "0"))
seqs, headers = synth.families(777, $N, 1000, length_jitter=j) if j else synth.families(777, $N, 1000)
synth.write_fasta("/tmp/cr_$N.fa", seqs, headers)
PY
s=$(date +%s.%N)
meshclust2_amd/host/msc_cluster /tmp/cr_$N.fa --recover $W --id 0.9 --kmer $K --datatype $DT --output /tmp/cr_one.clstr "$@" > /tmp/cr_one.log 2>&1
e=$(date +%s.%N); echo "one GPU: $(python3 -c "print(round($e - $s, 2))") s"
grep -E "timestamp (accumulate|update)" /tmp/cr_one.log | tr '\n' ' '; echo
s=$(date +%s.%N)
for r in $(seq 0 $((RANKS - 1))); do
  RANK=$r WORLD_SIZE=$RANKS LOCAL_RANK=$r MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 MSC_COMM=tcp MSC_ONE_GPU=1 \
    meshclust2_amd/host/msc_cluster /tmp/cr_$N.fa --recover $W --id 0.9 --kmer $K --datatype $DT --output /tmp/cr_ranks.clstr "$@" > /tmp/cr_ranks_$r.log 2>&1 &
done
wait
e=$(date +%s.%N); echo "$RANKS ranks on one GPU (sockets): $(python3 -c "print(round($e - $s, 2))") s"
grep -E "timestamp (accumulate|update)|collectives" /tmp/cr_ranks_0.log | cut -c1-300
cmp /tmp/cr_one.clstr /tmp/cr_ranks.clstr && echo "same .clstr bytes"
mkdir -p $R/gpurun_out/cr; cp /tmp/cr_one.clstr /tmp/cr_ranks.clstr /tmp/cr_one.log /tmp/cr_ranks_0.log $R/gpurun_out/cr/ 2>/dev/null
