# Builds everything in-tree for gfx950:
#   visual_slam_amd/liborbamd.so      HIP kernels + C ABI (include/orbamd.h)
#   build/orb_project<ext>.so         pybind11 module with the reference's name and location
#   oracle/liborb_oracle.so           CPU restatement (test infrastructure only)
HIPCC      ?= /opt/rocm/bin/hipcc
CXX        ?= g++
PYTHON     ?= python3
ARCH       ?= gfx950
CSRC       := visual_slam_amd/csrc
HIPFLAGS   := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result
EXT_SUFFIX := $(shell $(PYTHON)-config --extension-suffix 2>/dev/null || $(PYTHON) -c "import sysconfig;print(sysconfig.get_config_var('EXT_SUFFIX'))")
PYINC      := $(shell $(PYTHON) -m pybind11 --includes)

LIB    := visual_slam_amd/liborbamd.so
MODULE := build/orb_project$(EXT_SUFFIX)

all: $(LIB) $(MODULE) oracle trighost

$(LIB): $(CSRC)/orb_device.hip $(CSRC)/orb_capi.hip $(CSRC)/orb_internal.h $(CSRC)/orb_pattern.h $(CSRC)/orb_trig.h include/orbamd.h
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/orb_device.hip $(CSRC)/orb_capi.hip

$(MODULE): $(CSRC)/orb_module.cpp include/orbamd.h $(LIB)
	mkdir -p build
	$(CXX) -O2 -std=c++17 -fPIC -shared -fvisibility=hidden $(PYINC) $(CSRC)/orb_module.cpp -o $@ \
	    -Lvisual_slam_amd -lorbamd -Wl,-rpath,'$$ORIGIN/../visual_slam_amd'

oracle:
	$(MAKE) -s -C oracle

# Host build of csrc/orb_trig.h for tests/test_trig_kernels.py (test infrastructure; needs a CPU with FMA, which
# the test checks before loading it).
trighost: build/libtrig_host.so
build/libtrig_host.so: tests/trig_host.cpp $(CSRC)/orb_trig.h
	mkdir -p build
	$(CXX) -O2 -mfma -ffp-contract=off -shared -fPIC -o $@ tests/trig_host.cpp -lm

# Stand-alone measurement programs (not part of the library).
MICRO := build/stream_read build/valu_rates build/valu_peak build/mfma_fp4_probe build/mfma_valu_coexec build/mfma_top2_pattern build/coexec_classes build/mfma_port_block build/wave_simd_map
micro: $(MICRO)
build/%: $(CSRC)/micro/%.hip
	mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -Wno-unused-value -Wno-unused-result -o $@ $<

clean:
	rm -f $(LIB) $(MODULE) oracle/liborb_oracle.so oracle/liborb_oracle_O0.so

.PHONY: all oracle micro clean trighost
