#!/bin/bash
# Where do the waves of seg_slab_kernel spend their cycles?  One rocprofv3 --pmc pass of SQ counters per case (second launch reported).
#   bash tools/pmc_slab_sq.sh <out-dir>
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r04/pmc_slab_sq}
rm -rf "$OUT"; mkdir -p "$OUT"
C1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU"
C2="SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM"
for spec in "mh bfloat16" "mh float32" "gs128 float32" "gws bfloat16" "gws float32" "gs64 float32"; do
  set -- $spec
  for n in 1 2; do
    eval "C=\$C$n"
    rocprofv3 --pmc $C --output-format csv -d "$OUT/$1_$2_$n" -o pmc -- python3 tools/sweep_slab.py --case $1 --dtype $2 --seq 2:2 > "$OUT/$1_$2_$n.txt" 2> "$OUT/$1_$2_$n.err"
  done
done
find "$OUT" -type f ! -name "*.csv" ! -name "*.txt" ! -name "*.err" -delete
find "$OUT" -name "*.csv" -size +4M -delete
python3 tools/pmc_slab_sq.py "$OUT" | tee "$OUT/table.txt"
