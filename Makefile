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

clean:
	rm -rf build $(OUT)

.PHONY: all clean
