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

clean:
	rm -rf $(CSRC)/build $(LIB)

.PHONY: all clean
