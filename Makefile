# Builds libdclip_hip.so (gfx950 device code + C ABI) and the oracle's C pieces.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := dclip_amd/csrc
SRCS := $(wildcard $(CSRC)/*.hip)
OBJS := $(patsubst $(CSRC)/%.hip,$(CSRC)/build/%.o,$(SRCS))
LIB := dclip_amd/libdclip_hip.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -fvisibility=hidden -Wall -Wno-unused-function

all: $(LIB)

$(CSRC)/build/%.o: $(CSRC)/%.hip $(CSRC)/common.h include/dclip_hip.h
	@mkdir -p $(CSRC)/build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

# Diagnostic library with in-kernel time stamps in the GEMM (tools/gemm_stamps.py); never loaded by the product.
STAMPLIB := tools/ab/libdclip_hip_stamps.so
stamps: $(STAMPLIB)
$(STAMPLIB): $(OBJS) $(CSRC)/gemm_f32.hip $(CSRC)/gemm_bf16.hip
	@mkdir -p tools/ab
	$(HIPCC) $(HIPFLAGS) -DDCLIP_GEMM_STAMPS -c $(CSRC)/gemm_f32.hip -o $(CSRC)/build/gemm_f32_stamps.o
	$(HIPCC) $(HIPFLAGS) -DDCLIP_GEMM_STAMPS -c $(CSRC)/gemm_bf16.hip -o $(CSRC)/build/gemm_bf16_stamps.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(filter-out $(CSRC)/build/gemm_f32.o $(CSRC)/build/gemm_bf16.o,$(OBJS)) \
		$(CSRC)/build/gemm_f32_stamps.o $(CSRC)/build/gemm_bf16_stamps.o

clean:
	rm -rf $(CSRC)/build $(LIB)

.PHONY: all clean stamps
