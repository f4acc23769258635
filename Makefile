# Top-level build: the HIP library (the product), the developer harness, the CPU checkers.
#   make            -> geot_amd/libgeot_hip.so + tools/kbench
#   make oracle     -> oracle/libgeot_oracle.so   (test infrastructure)
#   make ref        -> oracle/_ref/*.so           (needs /root/reference; this container only)
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
HIPFLAGS = -O3 --offload-arch=$(ARCH) -std=c++17 -fPIC -Iinclude -Wno-unused-value

LIB = geot_amd/libgeot_hip.so
SRC = geot_amd/csrc/seg_reduce.hip geot_amd/csrc/seg_reduce_f32.hip geot_amd/csrc/seg_reduce_f64.hip geot_amd/csrc/seg_reduce_f16.hip \
      geot_amd/csrc/seg_reduce_bf16.hip geot_amd/csrc/seg_slab.hip geot_amd/csrc/seg_sort.hip geot_amd/csrc/seg_plan.hip \
      geot_amd/csrc/seg_guard.hip

.PHONY: all lib devlib tools shim oracle ref clean
all: lib tools

# one object per source under geot_amd/csrc/.obj, the stale ones compiled side by side, one link; staleness by content
# (geot_amd/_lib.py holds the recipe: hipcc $(HIPFLAGS) -c <src> -o <obj>, then hipcc -shared <objs> -o $(LIB))
lib: $(LIB)
$(LIB): $(SRC) include/geot_hip.h include/geot_hip_dev.h geot_amd/csrc/internal.h
	python3 geot_amd/_lib.py lib

# the DEVELOPMENT build: the same sources with -DGEOT_DEV_EXPERIMENTS (rejected kernel variants + the timing probe behind switches);
# loaded instead of the product by processes started with GEOT_HIP_LIB=dev (tools/, tests/test_gpu_dev_variants.py)
devlib:
	python3 geot_amd/_lib.py devlib

tools: tools/kbench
tools/kbench: tools/kbench.cpp $(LIB) include/geot_hip.h
	$(HIPCC) -O2 --offload-arch=$(ARCH) -std=c++17 -Iinclude tools/kbench.cpp -Lgeot_amd -lgeot_hip \
	  -Wl,-rpath,'$$ORIGIN/../geot_amd' -o $@

# the PyTorch dispatcher plugin over the C ABI (the host side of the drop-in): geot_amd/_C.so
shim: geot_amd/_C.so
# (geot_amd/_lib.py holds the recipe: g++ -c csrc/torch_ops.cpp csrc/host_*.cpp side by side, one link against libtorch + libgeot_hip)
geot_amd/_C.so: geot_amd/csrc/torch_ops.cpp geot_amd/csrc/host_state.cpp geot_amd/csrc/host_cache.cpp geot_amd/csrc/host_plan.cpp geot_amd/csrc/host.h $(LIB) include/geot_hip.h
	python3 geot_amd/_lib.py plugin

oracle:
	$(MAKE) -C oracle oracle
ref:
	$(MAKE) -C oracle ref

clean:
	rm -rf $(LIB) geot_amd/libgeot_hip_dev.so tools/kbench geot_amd/_C.so geot_amd/*.srchash geot_amd/csrc/.obj
	$(MAKE) -C oracle clean
