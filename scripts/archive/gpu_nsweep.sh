#!/bin/bash
# batch-size sweep N = 256 / 4096 / 16384 / 65536 chunks of 64 KiB (LZ4 and Snappy decode)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-ns}
mkdir -p "$OUT"; rm -f "$OUT/nsweep.jsonl"
for algo in lz4 snappy; do for mib in 16 256 1024 4096; do
  u=$(( mib < 64 ? mib : 64 ))
  timeout 400 python bench.py --algo $algo --mib-per-gpu $mib --unique-mib $u --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>> "$OUT/err.log" | tee -a "$OUT/nsweep.jsonl" | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['metric'], r['config']['chunks_per_gpu'], r['value'], r['roofline']['kernel_ms'])"
done; done
