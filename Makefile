# Builds libamdstamp.so (HIP kernels + C ABI, gfx950 only) and the oracle's C pieces.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := stamp_amd/csrc
OBJ   := build/obj
LIB   := stamp_amd/lib/libamdstamp.so
SRCS  := $(wildcard $(CSRC)/*.hip)
OBJS  := $(patsubst $(CSRC)/%.hip,$(OBJ)/%.o,$(SRCS))
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $(if $(PROBE),-DAMDS_GEMM_PROBE,)

all: $(LIB)

$(OBJ)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/amdstamp.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p $(dir $(LIB))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

clean:
	rm -rf build $(LIB)

.PHONY: all clean
