#!/bin/bash
# Developer loop for the graded kernel: compiles ONLY the headline instantiations (-DGEOT_HEADLINE_ONLY, ~15 s instead of
# minutes), prints their resource usage and leaves the gfx950 assembly in /tmp/geot_headline.s.
#   tools/isa.sh [extra hipcc flags]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -DGEOT_HEADLINE_ONLY -Wno-unused-value "$@" \
  --offload-device-only -S geot_amd/csrc/seg_reduce.hip -o /tmp/geot_headline.s -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Function Name|VGPRs:|SGPRs:|ScratchSize|Occupancy" | sed 's/.*remark: *//' | grep -A4 "seg_tile_kernel\|seg_fixup_kernel"
