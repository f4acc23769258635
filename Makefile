# Top-level build: the HIP library (the product), the developer harness, the CPU checkers.
#   make            -> geot_amd/libgeot_hip.so + tools/kbench
#   make oracle     -> oracle/libgeot_oracle.so   (test infrastructure)
#   make ref        -> oracle/_ref/*.so           (needs /root/reference; this container only)
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
HIPFLAGS = -O3 --offload-arch=$(ARCH) -std=c++17 -fPIC -Iinclude -Wno-unused-value

LIB = geot_amd/libgeot_hip.so
SRC = geot_amd/csrc/seg_reduce.hip

.PHONY: all lib tools oracle ref clean
all: lib tools

lib: $(LIB)
$(LIB): $(SRC) include/geot_hip.h
	$(HIPCC) $(HIPFLAGS) -shared $(SRC) -o $@

tools: tools/kbench
tools/kbench: tools/kbench.cpp $(LIB) include/geot_hip.h
	$(HIPCC) -O2 --offload-arch=$(ARCH) -std=c++17 -Iinclude tools/kbench.cpp -Lgeot_amd -lgeot_hip \
	  -Wl,-rpath,'$$ORIGIN/../geot_amd' -o $@

oracle:
	$(MAKE) -C oracle oracle
ref:
	$(MAKE) -C oracle ref

clean:
	rm -f $(LIB) tools/kbench
	$(MAKE) -C oracle clean
