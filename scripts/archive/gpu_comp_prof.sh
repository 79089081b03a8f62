#!/bin/bash
# phase clock of the LZ4 compressor (nvcomp_amd/lib/alt/libnvcomp_cprof.so = scripts/build_variants.sh cprof "-DNVCOMP_LZM_PROF")
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-cprof}
mkdir -p "$OUT"
for ds in ${DATASETS:-silesia_style}; do
  NVCOMP_AMD_LIB=$PWD/nvcomp_amd/lib/alt/libnvcomp_cprof.so timeout 200 python scripts/bench_roundtrip.py --algo lz4 --dataset $ds --unique-mib 32 --mib 1024 > "$OUT/$ds.json" 2> "$OUT/$ds.err"
  cat "$OUT/$ds.json"; grep phase_share "$OUT/$ds.err"
done
