#!/bin/bash
# Where do the waves of the source-blocked multi-head kernels spend their cycles - with the row gathers in place and with them DROPPED
# ("slab_probe": the table's buffer descriptor has zero records, the instruction stream is unchanged)?  rocprofv3 --pmc passes of SQ
# counters over tools/bench_slab_cases.py (bf16 H=4 x F=64 and fp32, Reddit-scale stand-in); the last launch of each kernel is reported.
#   bash tools/pmc_slab_probe.sh <out-dir> [dtypes]
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r05/pmc_slab_probe}
DT=${2:-bf16}
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > "$OUT/sq_counters_available.txt"
C1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU"
C2="SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM"
C3="SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED SQ_IFETCH SQ_WAIT_IFETCH"
for opt in "slab_pair=0" "slab_pair=0,slab_probe=1" "slab_pair=0,slab_probe=1,slab_window=-1" "slab_pair=1" "slab_pair=1,slab_probe=1"; do
  tag=$(echo "$opt" | tr ',=' '__')
  for n in 1 2 3; do
    eval "C=\$C$n"
    rocprofv3 --pmc $C --output-format csv -d "$OUT/${tag}_$n" -o pmc -- python3 tools/bench_slab_cases.py --only mh --dtypes $DT --iters 2 --options "$opt" > "$OUT/${tag}_$n.txt" 2> "$OUT/${tag}_$n.err"
  done
done
find "$OUT" -type f ! -name "*.csv" ! -name "*.txt" ! -name "*.err" -delete
find "$OUT" -name "*.csv" -size +4M -delete
python3 tools/pmc_slab_probe.py "$OUT" | tee "$OUT/table.txt"
