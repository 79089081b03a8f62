#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-abl}
mkdir -p "$OUT"
for v in a1 a2 window; do
  NVCOMP_AMD_LZ4_DECODE=$v timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-verify --unchecked > "$OUT/$v.json" 2> "$OUT/$v.err"
  python -c "
import json; r=json.load(open('$OUT/$v.json')); print('$v', r['value'], r['roofline']['kernel_ms'])"
done
