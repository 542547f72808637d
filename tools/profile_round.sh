#!/bin/bash
# Refreshes the raw material of profiles/ on the GPU box (one box, back to back), r05 form:
#   gpurun_out/prof/bench_plain.json   the default bench command, un-profiled
#   gpurun_out/prof/stats_piped        rocprofv3 --kernel-trace --stats of the default command (blocks on three streams: kernels stretch each other)
#   gpurun_out/prof/stats_one          the same with MSC_GEMM_NO_PIPE=1 (every kernel of a block on one stream: the per-kernel figures)
#   gpurun_out/prof/pmc_fetch, pmc_write, pmc_sq   separate --pmc passes of the one-stream command (never combined with a trace domain):
#                                      FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES
# then, in the build container:
#   python tools/summarize_profiles.py r05_allpairs gpurun_out/prof/stats_one gpurun_out/prof/pmc_fetch gpurun_out/prof/pmc_write "<profile_key of the bench line>" "" gpurun_out/prof/pmc_sq
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_plain.json 2> $O/bench_plain.err
echo "plain done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_piped -o stats -- python3 $R/bench.py --cpu-seconds 0 --no-secondary --steps 6 --warmup 2 > $O/bench_stats_piped.json 2> $O/bench_stats_piped.err
echo "piped stats done"
export MSC_GEMM_NO_PIPE=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_one -o stats -- python3 $R/bench.py --cpu-seconds 0 --no-secondary --steps 6 --warmup 2 > $O/bench_stats_one.json 2> $O/bench_stats_one.err
echo "one-stream stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o fetch -- python3 $R/bench.py --cpu-seconds 0 --no-secondary --steps 2 --warmup 1 > /dev/null 2> $O/pmc_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o write -- python3 $R/bench.py --cpu-seconds 0 --no-secondary --steps 2 --warmup 1 > /dev/null 2> $O/pmc_write.err
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $O/pmc_sq -o sq -- python3 $R/bench.py --cpu-seconds 0 --no-secondary --steps 2 --warmup 1 > /dev/null 2> $O/pmc_sq.err
echo "sq done"
# keep only the small CSVs
find $O/stats_piped $O/stats_one $O/pmc_fetch $O/pmc_write $O/pmc_sq -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -delete
