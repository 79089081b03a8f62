#!/bin/bash
# A/B build of ONE api/<name>_api.hip translation unit: nvcomp_amd/lib/alt/libnvcomp_<tag>.so (the other objects are the
# product's), from the working tree or, with REF=<commit>, from that commit's sources.
# usage: [REF=commit] build_api_variant.sh <name: cascaded|ans|bitcomp|deflate|lz4|snappy> <tag> [flags...]
set -e
cd "$(dirname "$0")/.."
name=$1; tag=$2; shift 2
make -s -C nvcomp_amd/csrc -j16 >/dev/null
mkdir -p nvcomp_amd/lib/alt /tmp/variants
SRC=$PWD
if [ -n "${REF:-}" ]; then
  rm -rf /tmp/refsrc_$tag && mkdir -p /tmp/refsrc_$tag
  git archive $REF nvcomp_amd/csrc include | tar -x -C /tmp/refsrc_$tag
  SRC=/tmp/refsrc_$tag
fi
OBJ=nvcomp_amd/lib/obj
REST=$(ls $OBJ/api/*.o $OBJ/hlif/*.o | grep -v "api/${name}_api.o")
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$SRC/include -I$SRC/nvcomp_amd/csrc -Wno-unused-function "$@" \
  -c $SRC/nvcomp_amd/csrc/api/${name}_api.hip -o /tmp/variants/${tag}_${name}_api.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o nvcomp_amd/lib/alt/libnvcomp_${tag}.so /tmp/variants/${tag}_${name}_api.o $REST
echo "built $tag"
