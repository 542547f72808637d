#!/bin/bash
# SQ counters of the long-list rank pass on cfg5's shape (tools/accumulate_profile.sh's set): instructions per launch of k_pair_ranks_items
set -e
NT=$1; PER=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/items_pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 - <<PY
import sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
from golden_util import cfg5_set
from meshclust2_amd import synth
seqs, hdrs = cfg5_set(n_templates=$NT, per_template=$PER, run_cap=3000)
synth.write_fasta("/tmp/acc_$NT.fa", seqs, hdrs)
PY
W=$R/tests/golden/weights_cfg5_u16_k9.txt
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O -o pmc -- $R/meshclust2_amd/host/msc_cluster /tmp/acc_$NT.fa --recover $W --id 0.6 --output /tmp/acc3.clstr --sparse > $O/run.log 2>&1
# HBM traffic, in passes of their own (the guide's rule: one counter set per run, never with a trace domain)
mkdir -p $O/fetch $O/write
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- $R/meshclust2_amd/host/msc_cluster /tmp/acc_$NT.fa --recover $W --id 0.6 --output /tmp/acc3.clstr --sparse > $O/run_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- $R/meshclust2_amd/host/msc_cluster /tmp/acc_$NT.fa --recover $W --id 0.6 --output /tmp/acc3.clstr --sparse > $O/run_write.log 2>&1
python3 - <<PY
import csv, glob, collections
for what in ("fetch", "write"):
    f = glob.glob("$O/%s/**/*counter_collection.csv" % what, recursive=True)[0]
    tot = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = next((x for x in ("k_pair_ranks_items", "k_rank_items_finish", "k_rank_pass_prep") if x in r["Kernel_Name"]), None)
        if not k: continue
        tot[k] += float(r["Counter_Value"]); n[k] += 1
    for k in tot:
        # (KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read: doubled, /opt/skills/guides/MI355X_MICROARCH.md, HBM)
        mb = tot[k] / n[k] * 1024 / 1e6 * (2 if what == "fetch" else 1)
        print(what, k, n[k], "launches,", "%.2f MB per launch%s" % (mb, " (counter x 2)" if what == "fetch" else ""))
PY
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    k = next((x for x in ("k_pair_ranks_items", "k_rank_items_finish", "k_rank_pass_prep", "k_pair_sparse_mp") if x in name), None)
    if not k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k in acc:
    print(k, n[k], {c: round(v / max(n[k], 1)) for c, v in acc[k].items()})
PY
head -2 $(ls $O/*counter_collection.csv | head -1) | cut -c1-400; find $O -type f ! -name "run.log" -delete
