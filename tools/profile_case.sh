#!/bin/bash
# One profiled configuration on the GPU box:  tools/profile_case.sh <tag> <sq: 0|1> <program and arguments ...>
#   gpurun_out/<tag>/plain.json        the command un-profiled (its JSON line, if it prints one)
#   gpurun_out/<tag>/stats             rocprofv3 --kernel-trace --stats
#   gpurun_out/<tag>/fetch, write      separate --pmc passes (never combined with a trace domain)
#   gpurun_out/<tag>/sq1, sq2          (sq = 1) SQ busy / wait / instruction-mix counters, 8 SQ slots per pass
# then folds them into gpurun_out/profiles/<tag>_kernel_stats.csv and <tag>_pmc_hbm.json (tools/summarize_profiles.py), which are
# copied into profiles/ (tracked) afterwards. The program itself follows rocprofv3's `--` (no env / bash -c hop: the profiler's
# preloaded library has initialised the GPU by then).
set -e
TAG=$1; SQ=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O $R/gpurun_out/profiles
cd /tmp && export TMPDIR=/tmp
# program paths are relative to the repository root
PROG=$1; shift
ARGS=()
for a in "$@"; do if [ -e "$R/$a" ]; then ARGS+=("$R/$a"); else ARGS+=("$a"); fi; done
$PROG "${ARGS[@]}" > $O/plain.json 2> $O/plain.err
echo "$TAG plain done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- $PROG "${ARGS[@]}" > $O/stats.json 2> $O/stats.err
echo "$TAG stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- $PROG "${ARGS[@]}" > /dev/null 2> $O/fetch.err
echo "$TAG fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- $PROG "${ARGS[@]}" > /dev/null 2> $O/write.err
echo "$TAG write done"
EXTRA=()
if [ "$SQ" = "1" ]; then
	rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq1 -o sq1 -- $PROG "${ARGS[@]}" > /dev/null 2> $O/sq1.err
	echo "$TAG sq1 done"
	rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/sq2 -o sq2 -- $PROG "${ARGS[@]}" > /dev/null 2> $O/sq2.err
	echo "$TAG sq2 done"
	EXTRA=($O/sq1 $O/sq2)
fi
KEY=$(python3 -c "import json,sys; print(json.loads(open('$O/plain.json').read().strip().splitlines()[-1])['roofline']['profile_key'])" 2>/dev/null || echo "")
python3 $R/tools/summarize_profiles.py $TAG $O/stats $O/fetch $O/write "$KEY" $R/gpurun_out/profiles "${EXTRA[@]}"
cp $O/plain.json $R/gpurun_out/profiles/${TAG}_bench.json
# keep only the small CSVs
find $O -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" ! -name "*.json" ! -name "*.err" -delete
