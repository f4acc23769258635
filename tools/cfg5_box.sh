#!/bin/bash
# One box's record for configs[4]'s per-GPU kernel: what the box is (memory / shader clocks as rocm-smi
# reports them, the driver's view of the device), what it can do (streamed-read ceiling, random-row rate pure and with the operator's
# write mix - geot_profile_box / geot_profile_box_rows on the 57 GB table itself), what the kernel does on it (bench.py's cfg5 leg),
# and the translation counters of that kernel (UTCL1 requests / misses, UTCL2 requests / misses, one rocprofv3 --pmc pass each).
# Run it on several boxes (one gpurun call each): bash tools/cfg5_box.sh <tag>   -> gpurun_out/r06/cfg5_boxes/<tag>/
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-box}
OUT=gpurun_out/r06/cfg5_boxes/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
{ hostname; date -u +%FT%TZ; /opt/rocm/bin/rocm-smi --showclocks --showmeminfo vram --showuniqueid --showperflevel 2>&1 | grep -v "^=\|^$"; } > "$OUT/box.txt" 2>&1
/opt/rocm/bin/rocminfo 2>/dev/null | grep -i "Marketing Name\|Compute Unit\|Max Clock\|Uuid" | head -12 >> "$OUT/box.txt"
python3 bench.py --full --steps 4 --warmup 2 --no-cpu-baseline --only-secondary cfg5 > "$OUT/bench_cfg5.json" 2> "$OUT/bench_cfg5.err"
python3 - "$OUT" <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/bench_cfg5.json"))
leg = d["secondary"]["gather_scatter_cfg5"]
r = leg["roofline"]
keys = ("box_read_ceiling_gbps", "box_sclk_mhz", "box_random_row_gbps_default_policy", "box_random_row_gbps_nt", "box_random_row_gbps_with_write_mix",
        "write_mix_run", "row_gather_gbps", "row_gather_frac_of_box_random_row", "row_gather_frac_of_box_row_mix", "moved_gbps")
print("kernel", leg["kernel"], "kernel_ms", round(leg["kernel_ms_rank0"], 3), "ms_per_step", round(leg["ms_per_step"], 3))
print(" ".join(f"{k}={r.get(k):.4g}" if isinstance(r.get(k), float) else f"{k}={r.get(k)}" for k in keys))
PY
for c in TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_HIT_sum TCC_MISS_sum; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -o pmc -- python3 bench.py --full --steps 2 --warmup 1 --no-cpu-baseline --only-secondary cfg5 \
      > "$OUT/pmc_$c.json" 2> "$OUT/pmc_$c.err"
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
for d in sorted(glob.glob(out + "/pmc_*")):
    if not os.path.isdir(d):
        continue
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f:
        print(os.path.basename(d), "no csv"); continue
    rows = [r for r in csv.DictReader(open(f[0])) if "seg_tile_kernel<float, 4, true, 0" in r["Kernel_Name"]]
    vals = [float(r["Counter_Value"]) for r in rows]
    print(f"{os.path.basename(d)[4:]:45s} dispatches {len(vals):2d}  mean {sum(vals) / max(len(vals), 1):.6g}")
PY
find "$OUT" -type f ! -name "*.csv" ! -name "*.json" ! -name "*.txt" ! -name "*.err" -delete
find "$OUT" -name "*.csv" -size +2M -delete
