# Builds libamdstamp.so (HIP kernels + C ABI, gfx950 only) and the oracle's C pieces.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := stamp_amd/csrc
OBJ   := build/obj
LIB   := stamp_amd/lib/libamdstamp.so
SRCS  := $(wildcard $(CSRC)/*.hip)
OBJS  := $(patsubst $(CSRC)/%.hip,$(OBJ)/%.o,$(SRCS))
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $(if $(PROBE),-DAMDS_GEMM_PROBE,) $(if $(GELU_DEG),-DAMDS_GELU_DEG=$(GELU_DEG),)

all: $(LIB)

$(OBJ)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/amdstamp.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# The token-major weight-gradient GEMM (gemm_4w16.h TN) reads its fragments with inline-asm ds_read_b64_tr_b16 the compiler cannot see as
# asynchronous: the build checks, on the assembly of THESE flags, that nothing touches a destination register before its s_waitcnt.
PYTHON ?= python3
TN_SRCS := gemm_bf16 gemm_f16
TN_OK   := $(patsubst %,$(OBJ)/%.tn_ok,$(TN_SRCS))
$(OBJ)/%.tn_ok: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/amdstamp.h tools/check_tn_isa.py
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) --cuda-device-only -S $< -o $(OBJ)/$*.s
	$(PYTHON) tools/check_tn_isa.py --asm $(OBJ)/$*.s
	@touch $@

$(LIB): $(OBJS) $(TN_OK)
	@mkdir -p $(dir $(LIB))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

clean:
	rm -rf build $(LIB)

.PHONY: all clean
