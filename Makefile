# Builds the gfx950 library (product) and the CPU oracle (test infrastructure).
#   make            -> raytracer_amd/librt_hip.so + oracle/liboracle.so + tools/rtbench
# hipcc cross-compiles for gfx950 without a GPU.  -ffp-contract=off on both sides: the reference's JIT never
# fuses a*b+c, and the bit-exact parity of K / hit ids depends on one rounding per operator.
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
ARCH     ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 --offload-arch=$(ARCH) -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -Wall -Wno-unused-function
ORCFLAGS ?= -O2 -std=c++17 -march=x86-64-v3 -ffp-contract=off -fno-fast-math -fPIC -Wall -pthread

CSRC := raytracer_amd/csrc
LIB  := raytracer_amd/librt_hip.so
ORC  := oracle/liboracle.so
BENCH := tools/rtbench
CALIB := tools/calib_fetch

all: $(LIB) $(ORC) $(BENCH) $(CALIB)

HDRS := $(CSRC)/rt_kernels.h $(CSRC)/rt_device.h $(CSRC)/rt_wide.h $(CSRC)/rt_ldsw.h $(CSRC)/rt_bvh.h include/rt_hip.h
OBJS := $(addprefix build/,rt_lib.o rt_lbvh.o rt_ploc.o rt_sah.o rt_wide.o rt_bvh.o rt_variant_spheres.o rt_variant_mesh.o rt_variant_boxes.o rt_variant_all.o)

build/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

build/rt_bvh.o: $(CSRC)/rt_bvh.cpp $(CSRC)/rt_bvh.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(LIB): $(OBJS)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(OBJS)

$(ORC): oracle/oracle.cpp include/rt_hip.h
	$(CXX) $(ORCFLAGS) -shared -o $@ oracle/oracle.cpp

# the C-ABI driver: plain g++, links the library and nothing else (no HIP headers, no Python)
$(BENCH): tools/rtbench.cpp include/rt_hip.h $(LIB)
	$(CXX) -O2 -std=c++17 -Wall -o $@ tools/rtbench.cpp -Lraytracer_amd -lrt_hip -Wl,-rpath,'$$ORIGIN/../raytracer_amd'

# FETCH_SIZE calibration for the traversal kernels' access pattern (profiles/r02_fetch_calibration.txt)
$(CALIB): tools/calib_fetch.hip
	$(HIPCC) -O3 --offload-arch=$(ARCH) -o $@ $<

clean:
	rm -rf $(LIB) $(ORC) $(BENCH) $(CALIB) build

.PHONY: all clean
