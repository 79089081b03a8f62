#!/bin/bash
# Round 4: Cascaded kernels without scratch (fresh lane ids) against round 3's, first-pass occupancy variants; pair kernels
# without their scratch slot array. usage: gpu_r4e.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-r4e}
mkdir -p "$OUT"
for v in cascold default casc7 casc8; do
  lib=$PWD/nvcomp_amd/lib/alt/libnvcomp_$v.so; [ $v = default ] && lib=$PWD/nvcomp_amd/lib/libnvcomp.so
  for ds in example_float_columns int32 mortgage_col0_like table; do
    o=$(NVCOMP_AMD_LIB=$lib timeout 200 python scripts/bench_roundtrip.py --algo cascaded --dataset $ds --unique-mib 32 --mib 1024 2>/dev/null | tail -1)
    echo "{\"lib\": \"$v\", \"dataset\": \"$ds\", \"line\": $o}" | tee -a "$OUT/cascaded_ab.jsonl"
  done
done
timeout 600 python scripts/ab_decode.py --libs nvcomp_amd/lib/alt/libnvcomp_pair.so nvcomp_amd/lib/alt/libnvcomp_team.so nvcomp_amd/lib/libnvcomp.so --cases mix16m,mix64m,mix128m,snappy64m --steps 5 --warmup 2 --out "$OUT/ab_pair.jsonl" 2>/dev/null | cut -c1-110
