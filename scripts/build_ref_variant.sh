#!/bin/bash
# A/B build of the LZ decoders as they were at <commit>: nvcomp_amd/lib/alt/libnvcomp_<tag>.so (the other objects are the
# current product's). usage: build_ref_variant.sh <commit> <tag> [extra -D flags]
set -e
cd "$(dirname "$0")/.."
commit=$1; tag=$2; shift 2
rm -rf /tmp/refsrc_$tag && mkdir -p /tmp/refsrc_$tag
git archive $commit nvcomp_amd/csrc include | tar -x -C /tmp/refsrc_$tag
OBJ=nvcomp_amd/lib/obj
REST=$(ls $OBJ/api/*.o $OBJ/hlif/*.o | grep -v "api/lz4_api.o\|api/snappy_api.o")
for f in lz4_api snappy_api; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I/tmp/refsrc_$tag/include -I/tmp/refsrc_$tag/nvcomp_amd/csrc -Wno-unused-function $* \
    -c /tmp/refsrc_$tag/nvcomp_amd/csrc/api/$f.hip -o /tmp/refsrc_$tag/$f.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o nvcomp_amd/lib/alt/libnvcomp_${tag}.so /tmp/refsrc_$tag/lz4_api.o /tmp/refsrc_$tag/snappy_api.o $REST
echo "built $tag from $commit"
