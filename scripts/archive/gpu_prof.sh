#!/bin/bash
# phase clocks of the LZ4 window decoder (-DNVCOMP_LZW_PROF build in nvcomp_amd/lib/prof/) on several datasets
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-prof}
mkdir -p "$OUT"
for ds in ${DATASETS:-mortgage_col0_like int32 zeros noise}; do
  NVCOMP_AMD_PROF=1 NVCOMP_AMD_LIB=$PWD/nvcomp_amd/lib/prof/libnvcomp_prof.so timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --mib-per-gpu 1024 --unique-mib 32 --dataset $ds --producer fast > "$OUT/prof_$ds.json" 2> "$OUT/prof_$ds.err"
  echo "$ds $(python -c "import json;print(json.load(open('$OUT/prof_$ds.json'))['value'])")"; tail -1 "$OUT/prof_$ds.err"
done
