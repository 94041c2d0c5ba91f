# Builds libsqoa_mi355x.so (gfx950 only) and the test oracle.
#   make            product library + oracle
#   make lib        product library only
#   make tools      C harnesses (sqoabench-style bench, sqoaconv-style converter)
HIPCC ?= /opt/rocm/bin/hipcc
CC ?= gcc
ARCH ?= gfx950
CSRC := seqoia_amd/csrc
OUT := seqoia_amd/lib
BUILD := build
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result
CFLAGS_HOST := -std=gnu99 -O2 -fPIC -Wall -Wextra

KERNELS := $(CSRC)/sqoa_encode.hip $(CSRC)/sqoa_decode.hip $(CSRC)/sqoa_decode_wave.hip $(CSRC)/sqoa_decode_bal.hip $(CSRC)/sqoa_decode_qoi.hip $(CSRC)/sqoa_strips.hip $(CSRC)/sqoa_runtime.hip $(CSRC)/sqoa_selftest.hip
# sqoa_encode_fb.o: sqoa_encode.hip once more, its QOI-compat kernels in the form that does not rest on the LDS's lane order (sqoa_selftest.hip)
OBJS := $(patsubst $(CSRC)/%.hip,$(BUILD)/%.o,$(KERNELS)) $(BUILD)/sqoa_encode_fb.o $(BUILD)/sqoa_host.o

all: lib oracle tools

lib: $(OUT)/libsqoa_mi355x.so

$(BUILD)/%.o: $(CSRC)/%.hip $(CSRC)/sqoa_common.h $(CSRC)/sqoa_decode_parts.h include/sqoa_hip.h include/seqoia.h
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(BUILD)/sqoa_encode_fb.o: $(CSRC)/sqoa_encode.hip $(CSRC)/sqoa_common.h include/sqoa_hip.h include/seqoia.h
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -DSQOA_LDS_ORDER_FB -c $< -o $@

$(BUILD)/sqoa_host.o: $(CSRC)/sqoa_host.c include/sqoa_hip.h include/seqoia.h
	@mkdir -p $(BUILD)
	$(CC) $(CFLAGS_HOST) -c $< -o $@

$(OUT)/libsqoa_mi355x.so: $(OBJS)
	@mkdir -p $(OUT)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -lpthread

oracle:
	$(MAKE) -s -C oracle

TOOLS := tools/bin/sqoabench_mi355x tools/bin/sqoaconv_mi355x tools/bin/sqoabatch_mi355x tools/bin/lds_atomic_order
tools: lib $(TOOLS)
tools/bin/%: tools/%.c include/seqoia.h include/sqoa_hip.h
	@mkdir -p tools/bin
	$(CC) -std=gnu99 -O2 -Wall -Iinclude $< -o $@ -L$(OUT) -lsqoa_mi355x -lpthread -Wl,-rpath,'$$ORIGIN/../../$(OUT)'

# the hardware property the QOI-compat encoder's probes rest on (sqoa_encode.hip, SQOA_ENC_Q_RTN): checked on the GPU box by tests/test_gpu_tools.py
tools/bin/lds_atomic_order: tools/ubench/lds_atomic_order.hip
	@mkdir -p tools/bin
	$(HIPCC) --offload-arch=$(ARCH) -O2 $< -o $@

clean:
	rm -rf $(BUILD) $(OUT)/*.so tools/bin

.PHONY: all lib oracle tools clean
