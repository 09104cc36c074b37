#!/bin/bash
# Timing ablations of the fused ResBlock step (csrc/resblock.hip: RVCX_PAIR_ABL): builds polgen-rvc_amd/librvcx_abl<N>.so
# for every N given (cross-compiles without a GPU), to be run on the GPU box as
#   for n in ...; do RVCX_LIBRARY=polgen-rvc_amd/librvcx_abl$n.so python tools/bench_pair.py 3 1; done
set -e
cd "$(dirname "$0")/.."
make -j8 >/dev/null
for n in "$@"; do
  mkdir -p build/abl$n
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -Wno-unused-result -DRVCX_PAIR_ABL=$n \
        -c polgen-rvc_amd/csrc/resblock.hip -o build/abl$n/resblock.o
  objs=$(ls build/*.o | grep -v '/resblock.o')
  hipcc --offload-arch=gfx950 -shared -fPIC $objs build/abl$n/resblock.o -o polgen-rvc_amd/librvcx_abl$n.so
  echo built polgen-rvc_amd/librvcx_abl$n.so
done
