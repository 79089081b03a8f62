#!/bin/bash
# A/B builds of the LZ compressors with compile-time tunables -> nvcomp_amd/lib/cab/libnvcomp_<tag>.so (only api/lz4_api.hip
# and api/snappy_api.hip are recompiled). Kept apart from lib/alt/, whose builds the decoder tests load.
# usage: build_comp_variants.sh tag1 "flags1" tag2 "flags2" ...
set -e
cd "$(dirname "$0")/.."
make -s -C nvcomp_amd/csrc -j16 >/dev/null
mkdir -p nvcomp_amd/lib/cab /tmp/cvariants
OBJ=nvcomp_amd/lib/obj
REST=$(ls $OBJ/api/*.o $OBJ/hlif/*.o | grep -v "api/lz4_api.o\|api/snappy_api.o")
build() { # tag, flags...
  local tag=$1; shift
  for f in lz4_api snappy_api; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Invcomp_amd/csrc -Wno-unused-function ${LZ_SCHED--mllvm -amdgpu-sched-strategy=max-ilp} "$@" \
      -c nvcomp_amd/csrc/api/$f.hip -o /tmp/cvariants/${tag}_$f.o 2>/tmp/cvariants/${tag}_$f.log &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o nvcomp_amd/lib/cab/libnvcomp_${tag}.so /tmp/cvariants/${tag}_lz4_api.o /tmp/cvariants/${tag}_snappy_api.o $REST
  grep -h "error" /tmp/cvariants/${tag}_*.log | sort | uniq -c | head -5
  echo "built $tag"
}
while [ $# -ge 2 ]; do
  build "$1" $2 &
  shift 2
done
wait
