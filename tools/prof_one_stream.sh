#!/bin/bash
# usage: prof_np.sh <tag> [ENV=VAL ...]  -- kernel stats of the default cfg2 step on one stream
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
MSC_GEMM_NO_PIPE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag/prof -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --queries 1024 --steps 10 --no-secondary --cpu-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/$tag/b.json 2>/dev/null
find $GRAFT_REPO_ROOT/gpurun_out/$tag/prof -type f ! -name "*kernel_stats.csv" -delete
