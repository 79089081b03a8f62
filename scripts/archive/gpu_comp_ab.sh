#!/bin/bash
# LZ compressors, default build + every A/B build under nvcomp_amd/lib/alt/: ratio, compress and decompress GB/s (1 GiB)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-cab}
mkdir -p "$OUT"
for lib in nvcomp_amd/lib/libnvcomp.so nvcomp_amd/lib/alt/libnvcomp_*.so; do
  for algo in ${ALGOS:-lz4 snappy}; do for ds in ${DATASETS:-silesia_style text int32}; do
    NVCOMP_AMD_LIB=$PWD/$lib timeout 200 python scripts/bench_roundtrip.py --algo $algo --dataset $ds --unique-mib 32 --mib 1024 2>> "$OUT/err.log" | tee -a "$OUT/roundtrip.jsonl" | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$lib'.split('/')[-1], r['algo'], r['dataset'], 'ratio', r['ratio'], 'comp', r['compress_GBps'], 'decomp', r['decompress_GBps'])"
  done; done
done
