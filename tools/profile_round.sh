#!/bin/bash
# Refreshes the raw material of profiles/ on the GPU box (one box, back to back):
#   gpurun_out/prof_stats  rocprofv3 --kernel-trace --stats of the default bench command
#   gpurun_out/pmc_fetch, gpurun_out/pmc_write   separate --pmc passes (never combined with a trace domain)
#   gpurun_out/bench_plain.json   the same command un-profiled
# then: python tools/summarize_profiles.py r01 gpurun_out/prof_stats gpurun_out/pmc_fetch gpurun_out/pmc_write "<config key>"
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
python3 $R/bench.py > $R/gpurun_out/bench_plain.json 2> $R/gpurun_out/bench_plain.err
echo "plain done"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o stats -- python3 $R/bench.py --cpu-seconds 0 > $R/gpurun_out/bench_stats.json 2> $R/gpurun_out/bench_stats.err
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o fetch -- python3 $R/bench.py --cpu-seconds 0 --steps 3 --warmup 1 > /dev/null 2> $R/gpurun_out/pmc_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o write -- python3 $R/bench.py --cpu-seconds 0 --steps 3 --warmup 1 > /dev/null 2> $R/gpurun_out/pmc_write.err
echo "write done"
# keep only the small CSVs
find $R/gpurun_out/prof_stats $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -delete
