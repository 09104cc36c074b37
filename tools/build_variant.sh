#!/bin/bash
# build polgen-rvc_amd/librvcx_<tag>.so with extra -D flags for ONE source file: build_variant.sh <tag> <file.hip> <flags...>
set -e
cd "$(dirname "$0")/.."
tag=$1; src=$2; shift 2
make -j8 >/dev/null
mkdir -p build/$tag
base=$(basename $src .hip)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -Wno-unused-result "$@" -c polgen-rvc_amd/csrc/$src -o build/$tag/$base.o
objs=$(for f in polgen-rvc_amd/csrc/*.hip; do b=$(basename $f .hip); [ "$b" != "$base" ] && echo build/$b.o; done)
hipcc --offload-arch=gfx950 -shared -fPIC $objs build/$tag/$base.o -o polgen-rvc_amd/librvcx_$tag.so
echo built polgen-rvc_amd/librvcx_$tag.so
