#!/bin/bash
# Round 6: SQ counters of the 16-bit multi-head SpMM at configs[3]'s graph - the matrix-core kernel in both forms (a wave per group /
# a pair of waves per group), with its gathers dropped, and the row-per-wave kernel it replaces.  Same recipe as tools/pmc_slab_probe.sh
# (three rocprofv3 --pmc passes per option set, the last launch of the kernel reported by tools/pmc_slab_probe.py).
#   bash tools/pmc_spmm_mfma.sh <out-dir>
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r06/pmc_spmm_mfma}
rm -rf "$OUT"; mkdir -p "$OUT"
C1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU"
C2="SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM"
C3="SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA"
for opt in ${OPTS:-"slab_spmm_mfma=1" "slab_spmm_mfma=0"}; do
  tag=$(echo "$opt" | tr ',=' '__')
  for n in 1 2 3; do
    eval "C=\$C$n"
    rocprofv3 --pmc $C --output-format csv -d "$OUT/${tag}_$n" -o pmc -- python3 tools/bench_slab_cases.py --only mh --dtypes bf16 --iters 2 --options "$opt" > "$OUT/${tag}_$n.txt" 2> "$OUT/${tag}_$n.err"
  done
done
find "$OUT" -type f ! -name "*.csv" ! -name "*.txt" ! -name "*.err" -delete
find "$OUT" -name "*.csv" -size +4M -delete
python3 tools/pmc_slab_probe.py "$OUT" | tee "$OUT/table.txt"
