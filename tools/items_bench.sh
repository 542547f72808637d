#!/bin/bash
# tools/items_bench.py plain and under rocprofv3 --kernel-trace --stats: wall per pass, then the kernels of a pass
#   tools/items_bench.sh <tag> [n_templates] [passes] [min_query_len]      -> gpurun_out/<tag>/
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/items_bench.py "$@" > $O/plain.log 2>&1
cat $O/plain.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- python3 $R/tools/items_bench.py "$@" > $O/stats.log 2>&1
python3 $R/tools/kstats.py $O/stats
