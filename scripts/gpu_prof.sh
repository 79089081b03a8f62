#!/bin/bash
# phase clocks of the window decoder (needs nvcomp_amd/lib/alt/libnvcomp_prof*.so from scripts/build_variants.sh)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-prof}
mkdir -p "$OUT"
for lib in nvcomp_amd/lib/alt/libnvcomp_prof*.so; do
  tag=$(basename $lib .so)
  NVCOMP_AMD_LIB=$PWD/$lib NVCOMP_AMD_PROF=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/$tag.json" 2> "$OUT/$tag.err"
  tail -1 "$OUT/$tag.err"
  python -c "
import json; r=json.load(open('$OUT/$tag.json')); print('$tag', r['value'], r['roofline']['kernel_ms'])"
done
