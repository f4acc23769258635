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
for c in TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_HIT_sum TCC_MISS_sum; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -o pmc -- python3 bench.py --full --steps 2 --warmup 1 --no-cpu-baseline --only-secondary cfg5 \
      > "$OUT/pmc_$c.json" 2> "$OUT/pmc_$c.err"
done
python3 tools/cfg5_box_summary.py "$OUT"
find "$OUT" -type f ! -name "*.csv" ! -name "*.json" ! -name "*.txt" ! -name "*.err" -delete
find "$OUT" -name "*.csv" -size +2M -delete
