# Builds the HIP library (gfx950) and the CPU oracle.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := pose_estimation_amd/csrc
LIB := pose_estimation_amd/libpose_amd.so
HDRS := $(wildcard $(CSRC)/*.h) include/pose_amd.h

SHIM := tests/host_shim/libhost_eig.so

FACADE_TEST := tests/cpp/facade_test
IO_TEST := tests/cpp/io_test

all: $(LIB) oracle $(SHIM) $(FACADE_TEST) $(IO_TEST)

$(LIB): $(CSRC)/pose_amd.hip $(HDRS)
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -ffp-contract=on -fPIC -shared -pthread -Wall -Wno-unused-function -o $@ $<

# Host build of the per-lane eigen step, used only by the CPU tests.
$(SHIM): tests/host_shim/eig_shim.hip $(HDRS)
	$(HIPCC) -O2 -std=c++17 --offload-arch=$(ARCH) -ffp-contract=on -fPIC -shared -Wno-unused-function -o $@ $<

# Test drivers of the C++ host facade (plain g++, link the library through its C ABI; the GPU
# one also links the CPU oracle, as its checker).
$(FACADE_TEST): tests/cpp/facade_test.cpp include/pose_amd.hpp include/pose_amd_types.hpp include/pose_amd_io.hpp include/pose_amd.h $(LIB) oracle
	g++ -std=c++17 -O1 -Wall -Iinclude $< -o $@ -Lpose_estimation_amd -lpose_amd -Loracle/build -loracle \
	    -Wl,-rpath,'$$ORIGIN/../../pose_estimation_amd' -Wl,-rpath,'$$ORIGIN/../../oracle/build'

$(IO_TEST): tests/cpp/io_test.cpp include/pose_amd.hpp include/pose_amd_types.hpp include/pose_amd_io.hpp include/pose_amd.h $(LIB)
	g++ -std=c++17 -O1 -Wall -Iinclude $< -o $@ -Lpose_estimation_amd -lpose_amd -Wl,-rpath,'$$ORIGIN/../../pose_estimation_amd'

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(LIB) $(SHIM) $(FACADE_TEST) $(IO_TEST)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
