#!/bin/bash
# Where one step of the step-serial accumulate stage goes: msc_cluster at BASELINE cfg5's shape (mixed lengths
# 500 b - 50 kb, k = 9, 16-bit, the reference's `--feat slow` model, --id 0.6) with
#   MSC_CLUSTER_PROFILE=1  the driver's own timers: window construction / get_close / mark + take / closest
#   MSC_PROFILE_CALLS=1    the library's timers inside the 1 x M call: slot list / launches / stream wait
# and once more under rocprofv3 --kernel-trace --stats (kernel time per step).
#   tools/accumulate_profile.sh <tag> <n_templates> <per_template> [msc_cluster flags ...]        -> gpurun_out/<tag>/
set -e
TAG=$1; NT=$2; PER=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 - <<PY
import sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
from golden_util import cfg5_set
from meshclust2_amd import synth
seqs, hdrs = cfg5_set(n_templates=$NT, per_template=$PER, run_cap=3000)
synth.write_fasta("/tmp/acc_$NT.fa", seqs, hdrs)
print("sequences", len(seqs), "bases", sum(len(s) for s in seqs))
PY
W=$R/tests/golden/weights_cfg5_u16_k9.txt
MSC_CLUSTER_PROFILE=1 MSC_PROFILE_CALLS=1 $R/meshclust2_amd/host/msc_cluster /tmp/acc_$NT.fa --recover $W --id 0.6 --output /tmp/acc.clstr "$@" > $O/plain.log 2>&1
grep -E "timestamp|Number of clusters|profile|\[msc\]" $O/plain.log
if [ -z "$ACC_NO_ROCPROF" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- $R/meshclust2_amd/host/msc_cluster /tmp/acc_$NT.fa --recover $W --id 0.6 --output /tmp/acc2.clstr "$@" > $O/stats.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open("$O/kernel_stats.csv", "w") as out:
    out.write("kernel,calls,total_ns,average_ns,percentage\n")
    for r in rows:
        name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        out.write('"%s",%s,%s,%s,%s\n' % (name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
print(open("$O/kernel_stats.csv").read()[:2500])
PY
find $O/stats -type f ! -name "*kernel_stats.csv" -delete
fi
