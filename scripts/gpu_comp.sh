#!/bin/bash
# compress-side check: round trips (ratio, compress / decompress GB/s) of the LZ codecs on several datasets
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-c}
mkdir -p "$OUT"; rm -f "$OUT/roundtrip.jsonl"
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -x -k "encode or snappy or lz4_decode" > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -2 "$OUT/pytest_gpu.log"
for algo in lz4 snappy; do for ds in silesia_style text table int32 zeros noise; do
  timeout 200 python scripts/bench_roundtrip.py --algo $algo --dataset $ds --unique-mib 32 --mib 1024 >> "$OUT/roundtrip.jsonl" 2>> "$OUT/roundtrip.err"
done; done
python -c "
import json
for l in open('$OUT/roundtrip.jsonl'):
    r=json.loads(l); print(r['algo'], r['dataset'], 'ratio', r['ratio'], 'comp', r['compress_GBps'], 'decomp', r['decompress_GBps'])"
tail -3 "$OUT/roundtrip.err"
