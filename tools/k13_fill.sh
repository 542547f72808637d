#!/bin/bash
# k_pair_sparse_mp over 8 000 equal 20 kb lists (k = 13) under the library's A/B switches, on one box (profiles/r03_notes.md)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
K13="python3 bench.py --cpu-seconds 0 --nseq 8000 --length 20000 --k 13 --dtype 64 --queries 8 --mode get_close --layout sparse --weights tests/golden/weights_k5_u16.txt"
run() { tag=$1; shift; env "$@" $K13 > gpurun_out/k13_$tag.json 2>gpurun_out/k13_fill.err || { tail -n 3 gpurun_out/k13_fill.err; return; }
  python3 -c "
import json
d=json.load(open('gpurun_out/k13_$tag.json')); print('k13 $tag', round(d['roofline']['avg_launch_ms'],4))"; }
run base A=1
run fill2 MSC_SPARSE_MP_FILL=2
run fill4 MSC_SPARSE_MP_FILL=4
run wide MSC_SPARSE_MP_CHUNK=575
run wide_fill2 MSC_SPARSE_MP_CHUNK=575 MSC_SPARSE_MP_FILL=2
run waves4 MSC_SPARSE_DIV_WAVES=4
run wide_waves4 MSC_SPARSE_MP_CHUNK=575 MSC_SPARSE_DIV_WAVES=4
