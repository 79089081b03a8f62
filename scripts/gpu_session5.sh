#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-s5}
mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest_gpu.log"
for ds in int32 float32 lowcard zeros noise; do
  timeout 200 python scripts/bench_roundtrip.py --algo cascaded --dataset $ds >> "$OUT/roundtrip.jsonl" 2>> "$OUT/roundtrip.err"
done
timeout 200 python scripts/bench_roundtrip.py --algo cascaded --dataset int32 --opts 4096,4,1,0,1 >> "$OUT/roundtrip.jsonl" 2>> "$OUT/roundtrip.err"
timeout 200 python scripts/bench_roundtrip.py --algo cascaded --dataset int32 --opts 4096,4,0,0,1 >> "$OUT/roundtrip.jsonl" 2>> "$OUT/roundtrip.err"
for algo in lz4 snappy; do for ds in silesia_style text int32; do
  timeout 200 python scripts/bench_roundtrip.py --algo $algo --dataset $ds --unique-mib 32 >> "$OUT/roundtrip.jsonl" 2>> "$OUT/roundtrip.err"
done; done
cat "$OUT/roundtrip.jsonl"; tail -3 "$OUT/roundtrip.err"
