#!/bin/bash
# VERDICT r2 item 7, candidate "fold the fix-up launch into the tile kernel": a COST PROBE, on one box.
#   A  product kernels (headline-only build):                      tile + fix-up
#   B  probe build (-DGEOT_EXP_HANDOFF), fix-up still launched:    tile' + fix-up      (results still right)
#   C  probe build, GEOT_EXP_NOFIX=1:                              tile' alone         (what a single launch would cost,
#      chains of two tiles only - hub chains would come on top; rows of longer chains are wrong in this run)
# Builds both libraries into tools/_ab/ when hipcc is there (the build container), then times them with tools/ab_libs.py.
cd "$(dirname "$0")/.."
mkdir -p tools/_ab gpurun_out/r03
FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -Iinclude -Wno-unused-value -DGEOT_HEADLINE_ONLY"
SRC="geot_amd/csrc/seg_reduce.hip geot_amd/csrc/seg_slab.hip geot_amd/csrc/seg_sort.hip geot_amd/csrc/seg_plan.hip"
if [ ! -f tools/_ab/libgeot_headline.so ] || [ geot_amd/csrc/seg_reduce.hip -nt tools/_ab/libgeot_headline.so ]; then
  /opt/rocm/bin/hipcc $FLAGS $SRC -o tools/_ab/libgeot_headline.so || exit 1
  /opt/rocm/bin/hipcc $FLAGS -DGEOT_EXP_HANDOFF $SRC -o tools/_ab/libgeot_handoff.so || exit 1
fi
if python3 -c "import torch,sys; sys.exit(0 if torch.cuda.is_available() else 1)"; then
  echo "== A vs B (fix-up launched in both)"; python3 tools/ab_libs.py tools/_ab/libgeot_headline.so tools/_ab/libgeot_handoff.so
  echo "== A vs C (probe build without the fix-up launch)"; GEOT_EXP_NOFIX=1 python3 tools/ab_libs.py tools/_ab/libgeot_headline.so tools/_ab/libgeot_handoff.so --no-check
fi
