#!/bin/bash
# GPU check of the Bitcomp / ANS codecs: parity tests + round-trip throughput.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-nc}
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_bitcomp.py tests/test_ans.py -m gpu -q --timeout 600 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu.log"
rm -f "$OUT/roundtrip.jsonl"
for ds in silesia_style text table float_csv int32 lowcard zeros noise; do
  timeout 200 python scripts/bench_roundtrip.py --algo ans --dataset $ds --unique-mib 32 --mib 1024 >> "$OUT/roundtrip.jsonl" 2>> "$OUT/roundtrip.err"
done
for spec in "int32 0,4" "int32 1,4" "float32 0,4" "silesia_style 0,1" "text 0,1" "zeros 0,1" "noise 0,1" "int32 0,6" "int32 0,2"; do
  set -- $spec
  timeout 200 python scripts/bench_roundtrip.py --algo bitcomp --dataset $1 --opts $2 --unique-mib 32 --mib 1024 >> "$OUT/roundtrip.jsonl" 2>> "$OUT/roundtrip.err"
done
python -c "
import json
for l in open('$OUT/roundtrip.jsonl'):
    r=json.loads(l); print(r['algo'], r['dataset'], r.get('opts'), 'ratio', r['ratio'], 'comp', r['compress_GBps'], 'decomp', r['decompress_GBps'])"
tail -3 "$OUT/roundtrip.err"
