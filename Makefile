# Builds the product library roki-fd_amd/librkfd_amd.so (host C + HIP for gfx950),
# the CPU oracle (test infrastructure) and the lane-emulator harness (test infrastructure).
HIPCC   ?= /opt/rocm/bin/hipcc
CC      ?= gcc
CXX     ?= g++
ARCH    ?= gfx950
PKG      = roki-fd_amd
CSRC     = $(PKG)/csrc
BUILD    = $(PKG)/build
INC      = -Iinclude -I$(CSRC) -I$(CSRC)/host -I$(BUILD)
CFLAGS   = -O2 -Wall -fPIC $(INC)
# machine LICM off: hoisted literals / addresses held across the step loop cost ~35 VGPRs and the third wave per SIMD
ROCM_LIBDIR ?= $(abspath $(dir $(realpath $(HIPCC)))/../lib)
HIPFLAGS = --offload-arch=$(ARCH) -O3 -fPIC $(INC) -Wno-unused-value -mllvm -disable-machine-licm -DRKFD_ROCM_LIBDIR='"$(ROCM_LIBDIR)"'

HOST_OBJS = $(BUILD)/rkfd_ztk.o $(BUILD)/rkfd_world.o $(BUILD)/rkfd_sim.o $(BUILD)/rkfd_devmodel.o
LIB = $(PKG)/librkfd_amd.so

all: $(LIB) oracle emu spec

$(BUILD):
	mkdir -p $(BUILD)

$(BUILD)/%.o: $(CSRC)/host/%.c include/*.h $(CSRC)/host/*.h | $(BUILD)
	$(CC) $(CFLAGS) -c $< -o $@

$(BUILD)/rkfd_devmodel.o: $(CSRC)/rkfd_devmodel.cpp $(CSRC)/*.h $(CSRC)/device/*.h include/*.h | $(BUILD)
	$(CXX) -std=c++17 $(CFLAGS) -c $< -o $@

# the compiler's per-kernel resource report is kept beside the object: tests/test_build_resources.py checks that no
# kernel spills to scratch and that all of them fit three waves per SIMD (168 VGPRs)
# the device sources as string literals inside the library: rkfdBatchSpecialize (hipRTC) needs no source files at run time
DEVSRC = include/rkfd_model.h $(CSRC)/rkfd_devmodel.h $(CSRC)/rkfd_device.h $(sort $(wildcard $(CSRC)/device/*.h))
$(BUILD)/rkfd_device_src.inc: $(DEVSRC) tools/embed_sources.py | $(BUILD)
	python3 tools/embed_sources.py $@ $(DEVSRC)

$(BUILD)/rkfd_capi.o: $(CSRC)/rkfd_capi.hip $(CSRC)/rkfd_capi_node.inc $(BUILD)/rkfd_device_src.inc $(CSRC)/*.h $(CSRC)/device/*.h include/*.h | $(BUILD)
	$(HIPCC) $(HIPFLAGS) -Rpass-analysis=kernel-resource-usage -c $< -o $@ 2> $(BUILD)/rkfd_capi.remarks || ( cat $(BUILD)/rkfd_capi.remarks; exit 1 )
	@grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|VGPRs Spill|SGPRs Spill" $(BUILD)/rkfd_capi.remarks | sed 's/.*remark: *//; s/ \[-Rpass.*//' > $(PKG)/kernel_resources.txt
	@grep -vE "remark:|^ +[0-9]+ \||^ +\||\^" $(BUILD)/rkfd_capi.remarks || true

$(LIB): $(HOST_OBJS) $(BUILD)/rkfd_capi.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -lm -ldl -lpthread

oracle:
	$(MAKE) -C oracle

# ahead-of-time specialised kernels: the code objects of the worlds of BASELINE.json's configurations, made now (hipRTC works
# without a GPU) and only LOADED at run time (rkfd_capi.hip: the store under $(PKG)/spec)
spec: $(LIB)
	python3 tools/make_spec.py

emu: tests/emu/librkfd_emu.so tests/emu/librkfd_emu_w2.so

tests/emu/librkfd_emu.so: tests/emu/rkfd_emu.cpp $(CSRC)/rkfd_device.h $(CSRC)/rkfd_devmodel.cpp $(CSRC)/*.h $(CSRC)/device/*.h include/*.h
	$(CXX) -std=c++20 -O2 -Wall -Wno-unknown-pragmas -fPIC -shared -pthread $(INC) -o $@ tests/emu/rkfd_emu.cpp $(CSRC)/rkfd_devmodel.cpp

# the same harness with two instances per wavefront (RKFD_W = 2, rkfd_devmodel.h)
tests/emu/librkfd_emu_w2.so: tests/emu/rkfd_emu.cpp $(CSRC)/rkfd_device.h $(CSRC)/rkfd_devmodel.cpp $(CSRC)/*.h $(CSRC)/device/*.h include/*.h
	$(CXX) -std=c++20 -O2 -Wall -Wno-unknown-pragmas -fPIC -shared -pthread -DRKFD_W=2 $(INC) -o $@ tests/emu/rkfd_emu.cpp $(CSRC)/rkfd_devmodel.cpp

clean:
	rm -rf $(BUILD) $(LIB) $(PKG)/spec tests/emu/librkfd_emu.so tests/emu/librkfd_emu_w2.so
	$(MAKE) -C oracle clean

.PHONY: all oracle emu spec clean
