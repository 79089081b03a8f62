#!/bin/bash
# A/B builds of the library with different compile-time tunables -> nvcomp_amd/lib/alt/libnvcomp_<tag>.so
# usage: build_variants.sh tag1 "flags1" tag2 "flags2" ...
set -e
cd "$(dirname "$0")/.."
mkdir -p nvcomp_amd/lib/alt
build() { # tag, flags...
  local tag=$1; shift
  local objs=""
  for f in nvcomp_amd/csrc/api/*.hip nvcomp_amd/csrc/hlif/*.hip; do
    o=/tmp/var_${tag}_$(basename $f .hip).o
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Invcomp_amd/csrc "$@" -c $f -o $o 2>/dev/null &
    objs="$objs $o"
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o nvcomp_amd/lib/alt/libnvcomp_${tag}.so $objs
  echo "built $tag"
}
while [ $# -ge 2 ]; do
  build "$1" $2
  shift 2
done
