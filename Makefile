# Builds the product library (HIP kernels + C ABI + C++ host mirror) for gfx950 and the CPU oracle.
# hipcc cross-compiles without a GPU.  Strict floating point everywhere: no contraction, no fast-math,
# correctly rounded f32 division and square root on the device.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    := gfx950   # the only target: the kernels use gfx950 instructions and its 160 KiB of LDS per CU
FPFLAGS := -ffp-contract=off -fno-fast-math
# no SLP vectorisation: packed f32 instructions issue at the rate of two scalar ones on gfx950, and pairing operands costs moves
DEVFLAGS:= --offload-arch=$(ARCH) -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize
CXXFLAGS:= -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wextra -Wno-unused-parameter $(FPFLAGS)

LIB     := rrt_amd/librrt_amd.so
OBJDIR  := build
OBJS    := $(OBJDIR)/kernels.o $(OBJDIR)/noalloc.o $(OBJDIR)/bvh_gpu.o $(OBJDIR)/capi.o $(OBJDIR)/multi.o $(OBJDIR)/scene_layout.o $(OBJDIR)/rrt.o $(OBJDIR)/host_capi.o

CLI     := rrt_amd/rrt

all: $(LIB) $(CLI) oracle

$(OBJDIR):
	mkdir -p $(OBJDIR)

$(OBJDIR)/kernels.o: rrt_amd/csrc/kernels.hip rrt_amd/csrc/kernels.h rrt_amd/csrc/scene_layout.h include/rrt_amd.h | $(OBJDIR)
	$(HIPCC) $(CXXFLAGS) $(DEVFLAGS) -c $< -o $@

$(OBJDIR)/noalloc.o: rrt_amd/csrc/noalloc.hip include/rrt_amd.h | $(OBJDIR)
	$(HIPCC) $(CXXFLAGS) $(DEVFLAGS) -c $< -o $@

$(OBJDIR)/bvh_gpu.o: rrt_amd/csrc/bvh_gpu.hip include/rrt_amd.h | $(OBJDIR)
	$(HIPCC) $(CXXFLAGS) $(DEVFLAGS) -Wno-unused-parameter -c $< -o $@

$(OBJDIR)/capi.o: rrt_amd/csrc/capi.cpp rrt_amd/csrc/kernels.h rrt_amd/csrc/scene_layout.h include/rrt_amd.h | $(OBJDIR)
	$(HIPCC) $(CXXFLAGS) -c $< -o $@

$(OBJDIR)/multi.o: rrt_amd/csrc/multi.cpp rrt_amd/csrc/kernels.h include/rrt_amd.h | $(OBJDIR)
	$(HIPCC) $(CXXFLAGS) -c $< -o $@

$(OBJDIR)/scene_layout.o: rrt_amd/csrc/scene_layout.cpp rrt_amd/csrc/scene_layout.h include/rrt_amd.h | $(OBJDIR)
	$(HIPCC) $(CXXFLAGS) -c $< -o $@

$(OBJDIR)/rrt.o: rrt_amd/host/rrt.cpp rrt_amd/host/rrt.hpp include/rrt_amd.h | $(OBJDIR)
	$(HIPCC) $(CXXFLAGS) -c $< -o $@

$(OBJDIR)/host_capi.o: rrt_amd/host/host_capi.cpp rrt_amd/host/rrt.hpp include/rrt_amd.h include/rrt_amd_host.h | $(OBJDIR)
	$(HIPCC) $(CXXFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(OBJS) -ldl -lpthread

$(OBJDIR)/rrt_cli.o: rrt_amd/host/rrt_cli.cpp rrt_amd/host/rrt.hpp include/rrt_amd.h include/rrt_amd_host.h | $(OBJDIR)
	$(HIPCC) $(CXXFLAGS) -c $< -o $@

# the C++ mirror is linked in as an object (the library exports only the C ABI)
$(CLI): $(OBJDIR)/rrt_cli.o $(OBJDIR)/rrt.o $(LIB)
	$(HIPCC) $(OBJDIR)/rrt_cli.o $(OBJDIR)/rrt.o -o $@ -Lrrt_amd -lrrt_amd -Wl,-rpath,'$$ORIGIN' -lpthread

oracle:
	$(MAKE) -C oracle

asm: rrt_amd/csrc/kernels.hip | $(OBJDIR)
	$(HIPCC) $(CXXFLAGS) $(DEVFLAGS) -S --cuda-device-only -Rpass-analysis=kernel-resource-usage $< -o $(OBJDIR)/kernels.s

# the library with per-wave busy-time counters in extend and shade (tools/wave_timing.sh); not part of `all`
wave-timing: $(LIB)
	$(HIPCC) $(CXXFLAGS) $(DEVFLAGS) -DRRT_WAVE_TIMING -c rrt_amd/csrc/kernels.hip -o $(OBJDIR)/kernels_wave_timing.o
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o rrt_amd/variant_wavetiming.so $(OBJDIR)/kernels_wave_timing.o $(filter-out $(OBJDIR)/kernels.o,$(OBJS)) -ldl -lpthread

clean:
	rm -rf $(OBJDIR) $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle asm wave-timing clean
