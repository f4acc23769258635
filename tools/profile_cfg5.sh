#!/bin/bash
# configs[4]'s per-GPU kernel under rocprofv3 (run on the MI355X box: `gpurun -- bash tools/profile_cfg5.sh`): kernel trace + stats,
# then one --pmc counter per pass (never combined with a trace domain; the program directly after `--`).  Lands in
# gpurun_out/prof_cfg5/; tools/derive_traffic.py picks the gather_scatter_cfg5 entry up from there.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_cfg5
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 -L > "$OUT/counters_avail.txt" 2>&1
w=gather_scatter_cfg5
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$w" -o bench -- python3 bench.py --full --steps 4 --warmup 2 --no-cpu-baseline --only-secondary cfg5 \
    > "$OUT/kt_$w.json" 2> "$OUT/kt_$w.err"
for c in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum ${EXTRA_PMC:-}; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_${c}__$w" -o pmc -- python3 bench.py --full --steps 2 --warmup 1 --no-cpu-baseline --only-secondary cfg5 \
      > "$OUT/pmc_${c}__$w.json" 2> "$OUT/pmc_${c}__$w.err"
done
find "$OUT" -type f ! -name "*.csv" ! -name "*.json" ! -name "*.txt" ! -name "*.err" -delete
find "$OUT" -name "*.csv" -size +4M -delete
du -sh "$OUT"; find "$OUT" -name "*.csv" | wc -l
tail -2 "$OUT"/*.err | grep -v "^$" | head -20
