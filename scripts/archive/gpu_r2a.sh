#!/bin/bash
# round 2, first GPU look at the two-kernel LZ4 decode: parity subset, indexed vs chase path, per-kernel times.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r2a}
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_lz4_decode.py tests/test_golden_decode.py -m gpu -q --timeout 300 -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -2 "$OUT/pytest.log"
B="python bench.py --no-cpu-baseline --no-extras"
for mib in 4096 1024 256; do
  for mb in 0 1000000000; do
    timeout 300 $B --steps 10 --warmup 2 --mib-per-gpu $mib --lz-index-min-batch $mb > "$OUT/lz4_${mib}_${mb}.json" 2> "$OUT/lz4_${mib}_${mb}.err"
    python -c "
import json; r=json.load(open('$OUT/lz4_${mib}_${mb}.json')); print('mib $mib min_batch $mb', r['value'], 'GB/s', r['roofline']['kernel_ms'], 'ms')"
  done
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o r -- python $OLDPWD/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --lz-index-min-batch 0 > "$OLDPWD/$OUT/prof.log" 2>&1
cd "$OLDPWD"
find "$OUT/prof" -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cp {} '"$OUT"'/kernel_stats.csv; head -8 {}'
find "$OUT" -name "*.csv" -size +4M -delete
