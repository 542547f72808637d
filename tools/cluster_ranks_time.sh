#!/bin/bash
# Wall time of the multi-rank driver (meshclust2_amd/cluster.py) next to the one-GPU C++ driver on the same file -- run on the GPU box
# (ranks share device 0 over gloo staged through host memory: the exchange logic, not xGMI).
#   tools/cluster_ranks_time.sh <n_seqs> <k> <dtype> <weights> <ranks>      (CLUSTER_TIME_JITTER=j: lengths 1000 +- j)
N=$1; K=$2; DT=$3; W=$4; RANKS=$5
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
meshclust2_amd/host/msc_cluster /tmp/cr_$N.fa --recover $W --id 0.9 --kmer $K --datatype $DT --output /tmp/cr_one.clstr > /tmp/cr_one.log 2>&1
e=$(date +%s.%N); echo "one GPU, C++ driver: $(python3 -c "print(round($e - $s, 2))") s"
s=$(date +%s.%N)
MSC_BENCH_BACKEND=gloo MSC_BENCH_ONE_GPU=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $RANKS --master-addr 127.0.0.1 --master-port 29611 -m meshclust2_amd.cluster /tmp/cr_$N.fa --recover $W --id 0.9 --kmer $K --datatype $DT --output /tmp/cr_ranks.clstr > /tmp/cr_ranks.log 2>&1
e=$(date +%s.%N); echo "$RANKS ranks on one GPU (gloo): $(python3 -c "print(round($e - $s, 2))") s"
tail -3 /tmp/cr_ranks.log | cut -c1-300
cmp /tmp/cr_one.clstr /tmp/cr_ranks.clstr && echo "same .clstr bytes"
mkdir -p $R/gpurun_out/cr; cp /tmp/cr_one.clstr /tmp/cr_ranks.clstr /tmp/cr_one.log /tmp/cr_ranks.log $R/gpurun_out/cr/ 2>/dev/null
