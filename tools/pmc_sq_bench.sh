#!/bin/bash
# SQ counters (where do the waves spend their cycles) of the kernels bench.py times: headline and chosen secondaries.
#   bash tools/pmc_sq_bench.sh <out-dir>
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r04/pmc_sq_bench}
rm -rf "$OUT"; mkdir -p "$OUT"
C1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU"
rocprofv3 --pmc $C1 --output-format csv -d "$OUT/headline" -o pmc -- python3 bench.py --full --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/headline.json" 2> "$OUT/headline.err"
for w in gws_cfg3 gws_cfg3_local gws_cfg3_bf16 mh_spmm_cfg4; do
  rocprofv3 --pmc $C1 --output-format csv -d "$OUT/$w" -o pmc -- python3 bench.py --full --steps 2 --warmup 1 --no-cpu-baseline --only-secondary $w > "$OUT/$w.json" 2> "$OUT/$w.err"
done
find "$OUT" -type f ! -name "*.csv" ! -name "*.json" ! -name "*.err" -delete
find "$OUT" -name "*.csv" -size +4M -delete
python3 - "$OUT" <<'P' | tee "$OUT/table.txt"
import csv, glob, os, sys
out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "*", ""))):
    name = os.path.basename(os.path.dirname(d))
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    per = {}
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "seg_" not in k:
            continue
        per.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        per[k].setdefault("ms", []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for k, v in per.items():
        m = {c: sum(x) / len(x) for c, x in v.items()}
        if m.get("ms", 0) < 0.2:
            continue
        wc = m.get("SQ_WAVE_CYCLES", 0) or 1
        print(f"{name}: {k[:90]}  {m['ms']:.3f} ms  parked {m.get('SQ_WAIT_ANY',0)/wc:.2f}  issue-stalled {m.get('SQ_WAIT_INST_ANY',0)/wc:.2f}  issuing {m.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f}"
              f"  VALU {m.get('SQ_INSTS_VALU',0):.3g}  SALU {m.get('SQ_INSTS_SALU',0):.3g}  LDS {m.get('SQ_INSTS_LDS',0):.3g}  VMEM_RD {m.get('SQ_INSTS_VMEM_RD',0):.3g}")
P
