#!/bin/bash
# Resource usage (VGPRs, occupancy) and gfx950 assembly of ANY instantiation of seg_tile_kernel, in seconds (the library takes
# minutes): tools/isa_inst.sh 'bf16_t, 8, false, 0, false, 3, RED_MEAN, 8' ['float, 4, true, 1, false, 0, RED_SUM, 8' ...]
# Template arguments as in csrc/seg_reduce.hip: <T, VEC, GATHER, WMODE, ATOMIC, NT, RED, U>.  Assembly: /tmp/geot_inst.s
cd "$(dirname "$0")/.."
src=/tmp/geot_inst.hip
{
  echo '#define GEOT_HEADLINE_ONLY'
  echo "#include \"$PWD/geot_amd/csrc/seg_reduce.hip\""
  echo 'namespace { void inst(SegParams p, hipStream_t st) {'
  for a in "$@"; do echo "  hipLaunchKernelGGL((seg_tile_kernel<$a>), dim3(1), dim3(kThreads), 0, st, p);"; done
  echo '} }'
  echo 'extern "C" void geot_dev_inst(void *st) { SegParams p{}; inst(p, static_cast<hipStream_t>(st)); }'
} > $src
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Igeot_amd/csrc -Wno-unused-value --offload-device-only -S $src -o /tmp/geot_inst.s \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | grep -A3 "seg_tile_kernel" | grep -v "^--"
