# Builds the HIP library (gfx950) and the CPU oracle.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := pose_estimation_amd/csrc
LIB := pose_estimation_amd/libpose_amd.so
HDRS := $(wildcard $(CSRC)/*.h) include/pose_amd.h

SHIM := tests/host_shim/libhost_eig.so

all: $(LIB) oracle $(SHIM)

$(LIB): $(CSRC)/pose_amd.hip $(HDRS)
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -Wall -Wno-unused-function -o $@ $<

# Host build of the per-lane eigen step, used only by the CPU tests.
$(SHIM): tests/host_shim/eig_shim.hip $(HDRS)
	$(HIPCC) -O2 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -Wno-unused-function -o $@ $<

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(LIB) $(SHIM)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
