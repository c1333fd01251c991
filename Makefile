# Builds the native pieces without Python (what __graft_entry__.build() does through _native.py):
#   make            libq2048_hip.so (hipcc, gfx950; cross-compiles without a GPU), libq2048_host.so (g++: the CPU twin),
#                   oracle/liboracle.so (gcc: the checker, test infrastructure only)
#   make examples   the two plain-C hosts of the ABI (examples/)
CSRC := 2048_q-learning_amd/csrc
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
CC ?= gcc
ROCM ?= /opt/rocm
DEPS := $(CSRC)/q2048_core.hpp $(CSRC)/q2048_core5.hpp $(CSRC)/q2048_luts.inc include/q2048.h

all: $(CSRC)/libq2048_hip.so $(CSRC)/libq2048_host.so oracle/liboracle.so

$(CSRC)/libq2048_hip.so: $(CSRC)/q2048_kernels.hip $(DEPS)
	$(HIPCC) -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -I include -I $(CSRC) -o $@ $<

$(CSRC)/libq2048_host.so: $(CSRC)/q2048_host.cpp $(DEPS)
	$(CXX) -O3 -std=c++17 -fPIC -shared -pthread -I include -I $(CSRC) -o $@ $<

oracle/liboracle.so:
	$(MAKE) -C oracle

examples: examples/rollout_host examples/rollout_host_cpu

examples/rollout_host: examples/rollout_host.c $(CSRC)/libq2048_hip.so
	$(CC) -std=c11 -O1 -D__HIP_PLATFORM_AMD__ -I $(ROCM)/include -I include $< -o $@ -L $(CSRC) -lq2048_hip \
	  -L $(ROCM)/lib -lamdhip64 -Wl,-rpath,$(abspath $(CSRC)) -Wl,-rpath,$(ROCM)/lib

examples/rollout_host_cpu: examples/rollout_host_cpu.c $(CSRC)/libq2048_host.so
	$(CC) -std=c11 -O1 -I include $< -o $@ -L $(CSRC) -lq2048_host -Wl,-rpath,$(abspath $(CSRC))

clean:
	rm -f $(CSRC)/libq2048_hip.so $(CSRC)/libq2048_host.so examples/rollout_host examples/rollout_host_cpu

.PHONY: all examples clean
