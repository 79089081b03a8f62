#!/bin/bash
# LZ decoders, default build + every A/B build under nvcomp_amd/lib/alt/, over the batch size
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-abn}
mkdir -p "$OUT"
for lib in nvcomp_amd/lib/libnvcomp.so nvcomp_amd/lib/alt/libnvcomp_*.so; do
  for algo in ${ALGOS:-lz4}; do for mib in ${MIBS:-256 1024 4096}; do
    u=$(( mib < 64 ? mib : 64 ))
    NVCOMP_AMD_LIB=$PWD/$lib timeout 300 python bench.py --algo $algo --mib-per-gpu $mib --unique-mib $u --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>> "$OUT/err.log" | tee -a "$OUT/lines.jsonl" | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$(basename $lib .so)', '$algo', r['config']['chunks_per_gpu'], r['value'])"
  done; done
done
