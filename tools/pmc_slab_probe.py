#!/usr/bin/env python3
"""Table of tools/pmc_slab_probe.sh: SQ counters of the LAST launch of each source-blocked kernel per option set (the plan-order
weight form of bench_slab_cases.py), per edge and as shares of the waves' cycles."""
import csv
import glob
import os
import sys

out = sys.argv[1]
NNZ = 114_615_892
for d in sorted(glob.glob(os.path.join(out, "*_1"))):
    name = os.path.basename(d)[:-2]
    vals, kern = {}, None
    for n in ("1", "2", "3"):
        f = glob.glob(os.path.join(out, name + "_" + n, "**", "*counter_collection.csv"), recursive=True)
        if not f:
            continue
        rows = [r for r in csv.DictReader(open(f[0])) if "seg_slab_w" in r["Kernel_Name"] or "seg_slab_kernel" in r["Kernel_Name"] or "seg_slab_spmm_mfma" in r["Kernel_Name"]]
        if any("seg_slab_spmm_mfma" in r["Kernel_Name"] for r in rows):      # (its gated vector-ALU twin is launched behind it and returns at once)
            rows = [r for r in rows if "seg_slab_spmm_mfma" in r["Kernel_Name"]]
        if not rows:
            continue
        last = max(int(r["Dispatch_Id"]) for r in rows)
        for r in rows:
            if int(r["Dispatch_Id"]) == last:
                vals[r["Counter_Name"]] = float(r["Counter_Value"])
                vals["ms_" + n] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                kern = r["Kernel_Name"][:60]
    wc = vals.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"{name}   [{kern}]")
    print("   " + " ".join(f"{k}={v:.4g}" for k, v in sorted(vals.items())))
    print("   shares of SQ_WAVE_CYCLES: parked (WAIT_ANY) %.2f, issue-stalled (WAIT_INST_ANY) %.2f, issuing (ACTIVE_INST_ANY) %.2f" %
          (vals.get("SQ_WAIT_ANY", 0) / wc, vals.get("SQ_WAIT_INST_ANY", 0) / wc, vals.get("SQ_ACTIVE_INST_ANY", 0) / wc))
    print("   wave-instructions per edge: " + " ".join(f"{k[9:]}={vals[k] / NNZ:.2f}" for k in sorted(vals) if k.startswith("SQ_INSTS_")))
