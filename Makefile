# Builds the HIP decode library (gfx950 only) and the CPU oracle used by the tests.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
PKG := parquet-java_amd
CSRC := $(PKG)/csrc
LIB := $(PKG)/lib/libpqdecode_hip.so
SRCS := $(CSRC)/pqd_api.hip $(CSRC)/pqd_chunk.hip $(CSRC)/k_unpack.hip $(CSRC)/k_rle.hip $(CSRC)/k_rle_plan.hip $(CSRC)/k_rle_small.hip $(CSRC)/k_rle_expand.hip $(CSRC)/k_dict_sliced.hip $(CSRC)/k_delta.hip $(CSRC)/k_bss.hip $(CSRC)/k_binary.hip $(CSRC)/k_binary_fused.hip $(CSRC)/k_plain.hip $(CSRC)/k_plain_binary.hip $(CSRC)/k_levels.hip $(CSRC)/k_delta_bytes.hip $(CSRC)/k_lz.hip $(CSRC)/k_zstd.hip $(CSRC)/k_gzip.hip
HDRS := include/pqd.h $(CSRC)/pqd_rle_scan.hpp $(CSRC)/pqd_device.hpp $(CSRC)/pqd_kernels.hpp $(CSRC)/pqd_rle_plan.hpp $(CSRC)/pqd_lz_common.hpp $(CSRC)/pqd_expand.hpp $(CSRC)/pqd_whole_page.hpp
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function

all: $(LIB) oracle cpptest

# one object per source (make -j builds them side by side; a change to one kernel file rebuilds that file only)
OBJS := $(patsubst $(CSRC)/%.hip,build/obj/%.o,$(SRCS))
build/obj/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build/obj
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(LIB): $(OBJS)
	@mkdir -p $(PKG)/lib
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(OBJS)

oracle:
	$(MAKE) -s -C oracle libpqd_oracle.so

# test program of the C++ reader mirror (include/pqd_readers.hpp); runs on the GPU box
CPPTEST := tests/cpp/test_readers
cpptest: $(CPPTEST)
$(CPPTEST): tests/cpp/test_readers.cpp include/pqd_readers.hpp include/pqd.h $(LIB)
	g++ -O1 -std=c++17 -Wall -Iinclude -o $@ $< -L$(PKG)/lib -lpqdecode_hip -Wl,-rpath,'$$ORIGIN/../../$(PKG)/lib' \
	    -Wl,-rpath-link,/opt/rocm/lib -Wl,--allow-shlib-undefined

asm: $(SRCS) $(HDRS)
	@mkdir -p build/asm
	cd build/asm && for f in k_unpack k_rle k_rle_plan k_rle_small k_rle_expand k_dict_sliced k_delta k_bss k_binary k_plain k_levels k_delta_bytes k_lz k_zstd; do \
	  $(HIPCC) $(HIPFLAGS) -S --cuda-device-only -o $$f.s ../../$(CSRC)/$$f.hip -Rpass-analysis=kernel-resource-usage 2> $$f.usage.txt; done

# development build with per-phase cycle counters in the sliced dictionary kernel (load it with PQD_LIB_PATH)
prof: $(SRCS) $(HDRS)
	@mkdir -p build/prof
	$(HIPCC) $(HIPFLAGS) -DPQD_PROF -shared -o build/prof/libpqdecode_hip.so $(SRCS)

clean:
	rm -rf $(PKG)/lib build $(CPPTEST)
	$(MAKE) -C oracle clean

.PHONY: all oracle asm prof clean cpptest

# development: variants of one kernel file built with -DBF_EXP=n etc. (make var VAR=1 VARFLAGS=-DBF_EXP=1 VARSRC=k_binary_fused)
var: $(OBJS)
	@mkdir -p build/var$(VAR)
	$(HIPCC) $(HIPFLAGS) $(VARFLAGS) -c -o build/var$(VAR)/$(VARSRC).o $(CSRC)/$(VARSRC).hip
	$(HIPCC) $(HIPFLAGS) -shared -o build/var$(VAR)/libpqdecode_hip.so $(filter-out build/obj/$(VARSRC).o,$(OBJS)) build/var$(VAR)/$(VARSRC).o
