# Builds libga_hip.so (gfx950 kernels + C ABI) in-tree.  hipcc cross-compiles without a GPU.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := guided-attention_amd/csrc
SRCS    := $(wildcard $(CSRC)/*.hip)
OBJS    := $(SRCS:.hip=.o)
LIB     := guided-attention_amd/libga_hip.so
# -amdgpu-mfma-vgpr-form: MFMA results stay in VGPRs (gfx950's register file is unified), which removes the
# v_accvgpr_read/write traffic around every softmax step (108 -> 0 per loop iteration in self_attn_fwd)
# -amdgpu-kernarg-preload-count=16: the first 16 dwords of scalar / pointer kernel arguments arrive in SGPRs at wave launch
# (gfx940+; the kernels keep a compatibility prologue that loads them when the firmware does not preload) — the scalar-load
# round trip in front of every kernel's first address computation goes away (all four passes -0.3 ... -0.9 % by the flag
# alone, profiles/r3_ab_kernarg_preload.txt; linear_kernel takes its tile geometry as 14 such dwords for it)
HIPFLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Iinclude -I$(CSRC) -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-kernarg-preload-count=16

all: $(LIB)

$(CSRC)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/ga_hip.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -Wl,--no-undefined -o $@ $(OBJS)

# Diagnostic libraries (never the product): linear.hip / conv3x3.hip with in-kernel clock stamps at their phase boundaries, the
# other objects as they are — tools/micro/lin_stamps.py and conv_stamps.py load them through GA_HIP_LIB.
STAMPLIB := tools/micro/libga_stamps.so
CSTAMPLIB := tools/micro/libga_conv_stamps.so
stamps: $(OBJS)
	$(HIPCC) $(HIPFLAGS) -DGA_LIN_STAMPS -c $(CSRC)/linear.hip -o tools/micro/linear_stamps.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -Wl,--no-undefined -o $(STAMPLIB) $(filter-out $(CSRC)/linear.o,$(OBJS)) tools/micro/linear_stamps.o
	$(HIPCC) $(HIPFLAGS) -DGA_CONV_STAMPS -c $(CSRC)/conv3x3.hip -o tools/micro/conv_stamps.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -Wl,--no-undefined -o $(CSTAMPLIB) $(filter-out $(CSRC)/conv3x3.o,$(OBJS)) tools/micro/conv_stamps.o

clean:
	rm -f $(OBJS) $(LIB) $(STAMPLIB) $(CSTAMPLIB) tools/micro/linear_stamps.o tools/micro/conv_stamps.o

.PHONY: all clean stamps
