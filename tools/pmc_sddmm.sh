#!/bin/bash
# seg_slab_sddmm_kernel under rocprofv3 --pmc: how many bytes do the 4-byte results written in ORIGINAL edge order cost?
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r04/pmc_sddmm}
rm -rf "$OUT"; mkdir -p "$OUT"
for c in WRITE_SIZE FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$n" -o pmc -- python3 tools/sweep_slab.py --case sddmm128 --seq 2:2 > "$OUT/$n.txt" 2> "$OUT/$n.err"
done
find "$OUT" -type f ! -name "*.csv" ! -name "*.txt" ! -name "*.err" -delete
find "$OUT" -name "*.csv" -size +4M -delete
python3 - "$OUT" <<'P' | tee "$OUT/table.txt"
import csv, glob, os, sys
out = sys.argv[1]
vals = {}
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "seg_slab_sddmm_kernel" in r["Kernel_Name"]]
    if not rows:
        continue
    last = max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"]) == last:
            vals[r["Counter_Name"]] = float(r["Counter_Value"])
            vals["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print("seg_slab_sddmm_kernel, 115 M edges, F=128 fp32:", " ".join(f"{k}={v:.4g}" for k, v in sorted(vals.items())))
if "WRITE_SIZE" in vals:
    print("written: %.2f GB (results: 0.46 GB)   fetched: %.2f GB (FETCH_SIZE x 2)" % (vals["WRITE_SIZE"] * 1024 / 1e9, vals.get("FETCH_SIZE", 0) * 2048 / 1e9))
P
