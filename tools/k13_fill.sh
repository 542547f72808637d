#!/bin/bash
# k_pair_sparse_mp over N equal 20 kb lists (k = 13), N = one fill of the 6 144-wave grid, 1.3 fills, two fills; both chunk sizes; one box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for n in 6144 8000 12288; do
for v in 512 575; do
  MSC_SPARSE_MP_CHUNK=$v python3 bench.py --cpu-seconds 0 --nseq $n --length 20000 --k 13 --dtype 64 --queries 8 --mode get_close --layout sparse --weights tests/golden/weights_k5_u16.txt --steps 8 > gpurun_out/k13_fill_${n}_$v.json 2>gpurun_out/k13_fill.err || { tail -n 3 gpurun_out/k13_fill.err; continue; }
  python3 -c "
import json
d=json.load(open('gpurun_out/k13_fill_${n}_$v.json')); print('lists $n chunk $v: %.4f ms per launch' % d['roofline']['avg_launch_ms'])"
done
done
