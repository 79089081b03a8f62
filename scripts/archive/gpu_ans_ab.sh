#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-ansab}
mkdir -p "$OUT"
for lib in nvcomp_amd/lib/libnvcomp.so nvcomp_amd/lib/alt/libnvcomp_ans*.so; do
  for ds in silesia_style text int32 zeros; do
    NVCOMP_AMD_LIB=$PWD/$lib timeout 200 python scripts/bench_roundtrip.py --algo ans --dataset $ds --unique-mib 32 --mib 1024 2>> "$OUT/err.log" | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$lib', r['dataset'], 'ratio', r['ratio'], 'comp', r['compress_GBps'], 'decomp', r['decompress_GBps'])"
  done
done
