#!/bin/bash
# SQ counters of k_pair_sparse_mp over 8 000 equal 20 kb lists (k = 13) at 512- and 575-entry chunks (one --pmc pass each, no trace domain)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in 512 575; do
  export MSC_SPARSE_MP_CHUNK=$v
  O=$R/gpurun_out/pmc_$v; rm -rf $O; mkdir -p $O
  timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq -o sq -- python3 $R/bench.py --cpu-seconds 0 --nseq 8000 --length 20000 --k 13 --dtype 64 --queries 8 --mode get_close --layout sparse --weights $R/tests/golden/weights_k5_u16.txt --steps 4 > $O/bench.json 2> $O/sq.err || exit 1
  python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for ff in glob.glob("$O/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(ff)):
        if "k_pair_sparse_mp" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("chunk $v: %.4f ms per launch under the profiler;" % d["roofline"]["avg_launch_ms"], ", ".join("%s %.4g" % (k, acc[k] / n[k]) for k in sorted(acc)))
PY
  find $O -type f ! -name "*.err" ! -name "bench.json" -delete
done
