# Builds the HIP library (gfx950) and the CPU oracle.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := pose_estimation_amd/csrc
LIB := pose_estimation_amd/libpose_amd.so
HDRS := $(wildcard $(CSRC)/*.h) include/pose_amd.h

all: $(LIB) oracle

$(LIB): $(CSRC)/pose_amd.hip $(HDRS)
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -Wall -Wno-unused-function -o $@ $<

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
