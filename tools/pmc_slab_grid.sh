#!/bin/bash
# L2 hit rate, fabric fetch bytes and duration of seg_slab_kernel per (slab size, lockstep window) at configs[3] scale:
# one rocprofv3 pass per counter (never combined with a trace), one kernel-trace pass; tools/pmc_slab_grid.py makes the table.
#   bash tools/pmc_slab_grid.sh <dtype> <out-dir> <seq> [extra sweep_slab.py flags, e.g. --coalesced]
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
DT=${1:-float32}; OUT=${2:-gpurun_out/r04/pmc_slab_$DT}; SEQ=${3:-2:0,2:1,2:2,2:3,1:1,1:2,1:4,4:0,4:1}; EXTRA=${4:-}
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -o t -- python3 tools/sweep_slab.py --case mh --dtype $DT --seq $SEQ $EXTRA > "$OUT/kt.txt" 2> "$OUT/kt.err"
for c in TCC_HIT_sum TCC_MISS_sum FETCH_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -o pmc -- python3 tools/sweep_slab.py --case mh --dtype $DT --seq $SEQ $EXTRA > "$OUT/pmc_$c.txt" 2> "$OUT/pmc_$c.err"
done
find "$OUT" -type f ! -name "*.csv" ! -name "*.txt" ! -name "*.err" -delete
find "$OUT" -name "*.csv" -size +4M -delete
python3 tools/pmc_slab_grid.py "$OUT" "$SEQ" | tee "$OUT/table.txt"
