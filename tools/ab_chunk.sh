#!/bin/bash
# same-box A/B of chunk sizes of k_pair_sparse_mp (libraries built ahead as meshclust2_amd/lib_<v>.so.tmp): the divergence form over
# 8 000 equal-length 20 kb sequences at k = 9, 11 and 13:   tools/ab_chunk.sh 512 575
# The built library is put back when the script ends, however it ends: later runs on the same box measure the tree, not the last variant.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cp meshclust2_amd/libmeshclust2_hip.so meshclust2_amd/libmeshclust2_hip.so.built
trap 'mv -f meshclust2_amd/libmeshclust2_hip.so.built meshclust2_amd/libmeshclust2_hip.so' EXIT
for v in "$@"; do
  cp meshclust2_amd/lib_$v.so.tmp meshclust2_amd/libmeshclust2_hip.so
  for k in 13; do
    python3 bench.py --cpu-seconds 0 --nseq 8000 --length 20000 --k $k --dtype 64 --queries 8 --mode get_close --layout sparse --weights tests/golden/weights_k5_u16.txt > gpurun_out/ab_k${k}_$v.json 2>gpurun_out/ab.err || { tail -n 3 gpurun_out/ab.err; continue; }
    python3 -c "
import json
d=json.load(open('gpurun_out/ab_k${k}_$v.json')); print('k=$k chunk $v', d['roofline']['kernel'], round(d['roofline']['avg_launch_ms'],4))"
  done
done
