#!/bin/bash
# A/B builds of the library with different LDS-window tunables -> nvcomp_amd/lib/alt/libnvcomp_<tag>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p nvcomp_amd/lib/alt
build() { # tag, flags...
  local tag=$1; shift
  local objs=""
  for f in nvcomp_amd/csrc/api/*.hip; do
    o=/tmp/var_${tag}_$(basename $f .hip).o
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Invcomp_amd/csrc "$@" -c $f -o $o &
    objs="$objs $o"
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o nvcomp_amd/lib/alt/libnvcomp_${tag}.so $objs
  echo "built $tag"
}
build w2k8  -DNVCOMP_LZW_OUTWIN=2048 -DNVCOMP_LZW_INRING=2048 -DNVCOMP_LZW_WAVES_PER_SIMD=8
build w3k7  -DNVCOMP_LZW_OUTWIN=3072 -DNVCOMP_LZW_BATCHMAX=1024 -DNVCOMP_LZW_KEEP=1792 -DNVCOMP_LZW_INRING=2048 -DNVCOMP_LZW_WAVES_PER_SIMD=7
build w4k6  -DNVCOMP_LZW_OUTWIN=4096 -DNVCOMP_LZW_BATCHMAX=1024 -DNVCOMP_LZW_KEEP=2816 -DNVCOMP_LZW_INRING=2048 -DNVCOMP_LZW_WAVES_PER_SIMD=6
build w6k4  -DNVCOMP_LZW_OUTWIN=6144 -DNVCOMP_LZW_BATCHMAX=1024 -DNVCOMP_LZW_KEEP=4864 -DNVCOMP_LZW_INRING=2048 -DNVCOMP_LZW_WAVES_PER_SIMD=4
build w2k8i1 -DNVCOMP_LZW_OUTWIN=2048 -DNVCOMP_LZW_INRING=1024 -DNVCOMP_LZW_WAVES_PER_SIMD=8
