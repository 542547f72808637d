#!/usr/bin/env python3
"""Folds rocprofv3 CSV output into the small per-round summaries committed under profiles/.

  python tools/summarize_profiles.py <tag> <stats_dir> <fetch_dir> <write_dir> [config key] [out dir] [extra pmc dirs ...]

writes <out dir>/<tag>_kernel_stats.csv and <out dir>/<tag>_pmc_hbm.json (out dir defaults to profiles/).
PMC correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE (KiB) reports exactly half of the bytes of a wide
coalesced streaming read, WRITE_SIZE is exact for 16-byte streaming stores -> hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
Extra pmc dirs (SQ_* passes): every counter found is averaged per kernel into "<kernel>": {"pmc": {name: mean per launch}}.
"""
import collections
import csv
import glob
import json
import os
import sys


def one(pattern):
    f = glob.glob(pattern, recursive=True)
    if not f:
        raise SystemExit("no file matches " + pattern)
    return f[0]


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0]


def main():
    tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
    key = sys.argv[5] if len(sys.argv) > 5 else ""
    out_dir = sys.argv[6] if len(sys.argv) > 6 and sys.argv[6] else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out_dir, exist_ok=True)
    rows = list(csv.DictReader(open(one(os.path.join(stats_dir, "**", "*kernel_stats.csv")))))
    with open(os.path.join(out_dir, "%s_kernel_stats.csv" % tag), "w") as f:
        f.write("kernel,calls,total_ns,average_ns,percentage,min_ns,max_ns\n")
        for r in rows:
            f.write("\"%s\",%s,%s,%s,%s,%s,%s\n" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]))
    pmc = {}
    for ctr, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(one(os.path.join(d, "**", "*counter_collection.csv")))):
            if r["Counter_Name"] == ctr:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            pmc.setdefault(k, {})[ctr] = {"launches": len(v), "mean_kib": sum(v) / len(v)}
    for k, d in pmc.items():
        if isinstance(d, dict) and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_bytes_per_launch_corrected"] = (2 * d["FETCH_SIZE"]["mean_kib"] + d["WRITE_SIZE"]["mean_kib"]) * 1024
    for d in sys.argv[7:]:
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(files[0])):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, ctrs in agg.items():
            for c, v in ctrs.items():
                pmc.setdefault(k, {}).setdefault("pmc", {})[c] = sum(v) / len(v)
    # VALU-busy fraction as the gfx94x derived metric states it (ROCm 7.2 ships no gfx950 section): cycles with a VALU instruction in
    # flight x 4 (a wave64 instruction occupies its SIMD for four) / SIMDs / elapsed cycles; elapsed = SQ_BUSY_CYCLES / 32 (the
    # counter sums the 8 XCDs x 4 shader engines). On the r02 digest kernel this gives 0.84 where instruction counts x measured
    # cycle costs gave 0.86.
    for k, d in pmc.items():
        c = d.get("pmc") if isinstance(d, dict) else None
        if c and c.get("SQ_BUSY_CYCLES") and "SQ_ACTIVE_INST_VALU" in c:
            d["valu_busy"] = c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (c["SQ_BUSY_CYCLES"] / 32)
        # matrix-pipe busy fraction: SQ_VALU_MFMA_BUSY_CYCLES counts the cycles a SIMD's matrix pipe is busy, summed over the chip's 1 024
        # SIMDs (the gfx94x MfmaUtil expression: / (elapsed cycles x SIMDs)); elapsed as above. Cross-check beside it: MFMA instructions
        # x 32 cycles (v_mfma_f32_32x32x64_f8f6f4 with FP4 operands, tools/ubench/mfma_rate.hip) / the same denominator.
        if c and c.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            elapsed = c["SQ_BUSY_CYCLES"] / 32
            d["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / elapsed
            if "SQ_INSTS_MFMA" in c:
                d["mfma_busy_from_instruction_count"] = c["SQ_INSTS_MFMA"] * 32 / 1024 / elapsed
    pmc["_config"] = key
    json.dump(pmc, open(os.path.join(out_dir, "%s_pmc_hbm.json" % tag), "w"), indent=1, sort_keys=True)
    print("wrote %s/%s_kernel_stats.csv and %s_pmc_hbm.json" % (out_dir, tag, tag))


if __name__ == "__main__":
    main()
