# Builds libga_hip.so (gfx950 kernels + C ABI) in-tree.  hipcc cross-compiles without a GPU.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := guided-attention_amd/csrc
SRCS    := $(wildcard $(CSRC)/*.hip)
OBJS    := $(SRCS:.hip=.o)
LIB     := guided-attention_amd/libga_hip.so
# -amdgpu-mfma-vgpr-form: MFMA results stay in VGPRs (gfx950's register file is unified), which removes the
# v_accvgpr_read/write traffic around every softmax step (108 -> 0 per loop iteration in self_attn_fwd)
HIPFLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Iinclude -I$(CSRC) -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form

all: $(LIB)

$(CSRC)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/ga_hip.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -Wl,--no-undefined -o $@ $(OBJS)

clean:
	rm -f $(OBJS) $(LIB)

.PHONY: all clean
