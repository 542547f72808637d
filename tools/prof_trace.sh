#!/bin/bash
# usage: prof_trace.sh <tag> [ENV=VAL ...]  -- kernel trace (start / end per dispatch) of a short cfg2 run, kept as CSV
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --queries 1024 --steps 3 --warmup 1 --no-secondary --cpu-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/$tag/tb.json 2>/dev/null
find $GRAFT_REPO_ROOT/gpurun_out/$tag/trace -type f ! -name "*kernel_trace.csv" -delete
