#!/bin/bash
# DEFLATE decoder, default build + every A/B build under nvcomp_amd/lib/alt/: bench.py --algo deflate
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-dab}
mkdir -p "$OUT"
for lib in nvcomp_amd/lib/libnvcomp.so nvcomp_amd/lib/alt/libnvcomp_*.so; do
  for mib in ${MIBS:-1024 4096}; do
    tag=$(basename $lib .so)_$mib
    NVCOMP_AMD_LIB=$PWD/$lib timeout 300 python bench.py --algo deflate --mib-per-gpu $mib --unique-mib 32 --no-cpu-baseline ${EXTRA:-} > "$OUT/$tag.json" 2>> "$OUT/err.log"
    python -c "
import json; r=json.load(open('$OUT/$tag.json')); print('$tag', r['value'], r['roofline']['kernel_ms'])"
  done
done
