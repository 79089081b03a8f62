#!/bin/bash
# two waves per chunk (producer / consumer) vs one wave per chunk over the batch size, HC mix; ALGOS default "lz4 snappy"
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-pair}
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_lz4_decode.py tests/test_snappy.py tests/test_fuzz_decode.py tests/test_fuzz_corrupt.py tests/test_golden_decode.py -m gpu -q -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -2 "$OUT/pytest.log"
for algo in ${ALGOS:-lz4 snappy}; do for mib in ${MIBS:-16 64 128 256 512}; do for pm in 0 1000000000; do
  u=$(( mib < 64 ? mib : 64 ))
  timeout 300 python bench.py --algo $algo --no-cpu-baseline --no-extras --mib-per-gpu $mib --unique-mib $u --steps 20 --warmup 3 --lz-pair-max-batch $pm 2>> "$OUT/err.log" | tee -a "$OUT/pair_sweep.jsonl" | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$algo chunks', r['config']['chunks_per_gpu'], 'pair_max $pm', r['value'], 'GB/s', r['roofline']['kernel_ms'], 'ms')"
done; done; done
