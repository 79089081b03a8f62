#!/bin/bash
# A/B build of the Cascaded codec with compile-time flags -> nvcomp_amd/lib/cab/libnvcomp_<tag>.so (only api/cascaded_api.hip
# is recompiled). usage: build_casc_variant.sh tag "flags"
set -e
cd "$(dirname "$0")/.."
make -s -C nvcomp_amd/csrc -j16 >/dev/null
mkdir -p nvcomp_amd/lib/cab /tmp/cvariants
OBJ=nvcomp_amd/lib/obj
REST=$(ls $OBJ/api/*.o $OBJ/hlif/*.o | grep -v "api/cascaded_api.o")
tag=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Invcomp_amd/csrc -Wno-unused-function $* \
  -c nvcomp_amd/csrc/api/cascaded_api.hip -o /tmp/cvariants/${tag}_cascaded_api.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o nvcomp_amd/lib/cab/libnvcomp_${tag}.so /tmp/cvariants/${tag}_cascaded_api.o $REST
echo "built $tag"
