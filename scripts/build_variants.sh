#!/bin/bash
# A/B builds of the LZ decoders with different compile-time tunables -> nvcomp_amd/lib/alt/libnvcomp_<tag>.so
# (only api/lz4_api.hip and api/snappy_api.hip are recompiled; everything else is the product's objects)
# usage: build_variants.sh tag1 "flags1" tag2 "flags2" ...
set -e
cd "$(dirname "$0")/.."
make -s -C nvcomp_amd/csrc -j16 >/dev/null
mkdir -p nvcomp_amd/lib/alt /tmp/variants
OBJ=nvcomp_amd/lib/obj
REST=$(ls $OBJ/api/*.o $OBJ/hlif/*.o | grep -v "api/lz4_api.o\|api/snappy_api.o")
build() { # tag, flags...
  local tag=$1; shift
  for f in lz4_api snappy_api; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Invcomp_amd/csrc -Wno-unused-function ${LZ_SCHED--mllvm -amdgpu-sched-strategy=max-ilp} "$@" \
      -c nvcomp_amd/csrc/api/$f.hip -o /tmp/variants/${tag}_$f.o 2>/tmp/variants/${tag}_$f.log &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o nvcomp_amd/lib/alt/libnvcomp_${tag}.so /tmp/variants/${tag}_lz4_api.o /tmp/variants/${tag}_snappy_api.o $REST
  grep -h "warning\|error" /tmp/variants/${tag}_*.log | sort | uniq -c | head -5
  echo "built $tag"
}
while [ $# -ge 2 ]; do
  build "$1" $2
  shift 2
done
