#!/bin/bash
# bench every A/B build under nvcomp_amd/lib/alt/ (scripts/build_variants.sh) on the LZ4 headline (verified output)
# usage: gpu_ab.sh <tag> ; env ALGOS (default lz4), MIBS (default "4096 1024")
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-ab}
mkdir -p "$OUT"
for lib in nvcomp_amd/lib/alt/libnvcomp_*.so; do
  tag=$(basename $lib .so)
  for algo in ${ALGOS:-lz4}; do
    for mib in ${MIBS:-4096 1024}; do
      NVCOMP_AMD_LIB=$PWD/$lib timeout 300 python bench.py --algo $algo --steps 5 --warmup 1 --no-cpu-baseline --no-extras --mib-per-gpu $mib --lz-index-min-batch 1000000000 > "$OUT/${tag}_${algo}_$mib.json" 2> "$OUT/${tag}_${algo}_$mib.err"
      python - "$OUT/${tag}_${algo}_$mib.json" <<'PY'
import json,sys
f=sys.argv[1]
try:
    r=json.load(open(f)); print(f.split('/')[-1], r['value'], 'GB/s', r['roofline']['kernel_ms'], 'ms')
except Exception as e: print(f,'ERR',e)
PY
    done
  done
done
