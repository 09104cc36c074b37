# Builds librvcx.so (HIP, gfx950 only) in-tree.  `python -c "import __graft_entry__ as g; g.build()"`
# drives this; the .so travels to the GPU box with the repo snapshot.
HIPCC ?= hipcc
ARCH  ?= gfx950
CSRC  := polgen-rvc_amd/csrc
OUT   := polgen-rvc_amd/librvcx.so
SRCS  := $(wildcard $(CSRC)/*.hip)
OBJS  := $(patsubst $(CSRC)/%.hip,build/%.o,$(SRCS))
HDRS  := $(wildcard $(CSRC)/*.h) include/rvcx.h
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wno-pass-failed -Wno-unused-result $(EXTRA)

all: $(OUT)

build/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(FLAGS) -c $< -o $@

$(OUT): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -o $@

# ---- host-side sanitizer build (CPU container only; never on the GPU box) ------------------------------------------
# The HOST pass of every .hip file under AddressSanitizer + UBSan, linked against tools/hipstub (HIP runtime on host memory,
# kernel launches are no-ops): build/asan/librvcx_asan.so.  tools/host_asan.sh builds it and runs tools/host_asan_driver.py.
CLANGXX ?= /opt/rocm/lib/llvm/bin/clang++
ASAN_FLAGS := --cuda-host-only -O1 -g -std=c++17 -fPIC -Wno-pass-failed -Wno-unused-result -fsanitize=address,undefined \
              -fno-sanitize-recover=undefined -fno-omit-frame-pointer
ASAN_OBJS := $(patsubst $(CSRC)/%.hip,build/asan/%.o,$(SRCS))

build/asan/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build/asan
	$(HIPCC) $(ASAN_FLAGS) -c $< -o $@

build/asan/hipstub.o: tools/hipstub/hipstub.cpp
	@mkdir -p build/asan
	$(HIPCC) $(ASAN_FLAGS) -x hip -c $< -o $@

# one dummy definition per translation unit's __hip_fatbin_<hash> (the device code objects a host-only pass does not have)
build/asan/fatbins.c: $(ASAN_OBJS)
	nm $(ASAN_OBJS) | awk '/ U __hip_fatbin_/ {print "char " $$2 "[64] = {0};"}' | sort -u > $@

build/asan/librvcx_asan.so: $(ASAN_OBJS) build/asan/hipstub.o build/asan/fatbins.c
	$(CLANGXX) -shared -fPIC -fsanitize=address,undefined -shared-libasan $(ASAN_OBJS) build/asan/hipstub.o build/asan/fatbins.c -o $@

host-asan: build/asan/librvcx_asan.so

clean:
	rm -rf build $(OUT)

.PHONY: all clean host-asan
