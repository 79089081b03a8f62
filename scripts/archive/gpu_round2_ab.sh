#!/bin/bash
# On the GPU box: the default library and every nvcomp_amd/lib/alt/libnvcomp_*.so through the headline decode bench
# (LZ4, Snappy; 4 GiB, verified) and the LZ round trip (compress GB/s + ratio). One line per library and leg.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r2ab}
mkdir -p "$OUT"
for lib in nvcomp_amd/lib/libnvcomp.so nvcomp_amd/lib/alt/libnvcomp_*.so; do
  [ -f "$lib" ] || continue
  for a in lz4 snappy; do
    NVCOMP_AMD_LIB=$PWD/$lib timeout 200 python bench.py --algo $a --steps 10 --no-cpu-baseline --no-extras 2>> "$OUT/err.log" | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$lib', '$a', 'decode', r['value'], 'kernel_ms', r['roofline']['kernel_ms'], 'verified', r['config']['verified'])" | tee -a "$OUT/ab.log"
    NVCOMP_AMD_LIB=$PWD/$lib timeout 200 python scripts/bench_roundtrip.py --algo $a --dataset silesia_style --unique-mib 32 --mib 1024 2>> "$OUT/err.log" | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$lib', '$a', 'roundtrip ratio', r['ratio'], 'comp', r['compress_GBps'], 'decomp', r['decompress_GBps'])" | tee -a "$OUT/ab.log"
  done
done
