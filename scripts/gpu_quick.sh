#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-q}
mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -x -k "not programs" > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -2 "$OUT/pytest_gpu.log"
for v in a1 a2 window; do
  NVCOMP_AMD_LZ4_DECODE=$v timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-verify --unchecked > "$OUT/$v.json" 2> "$OUT/$v.err"
  python -c "
import json; r=json.load(open('$OUT/$v.json')); print('$v', r['value'], r['roofline']['kernel_ms'])"
done
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$OUT/lz4.json" 2> "$OUT/lz4.err"
timeout 300 python bench.py --algo snappy --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$OUT/snappy.json" 2> "$OUT/snappy.err"
python -c "
import json
for f in ['lz4','snappy']:
    r=json.load(open('$OUT/'+f+'.json')); print(f, r['value'], r['roofline']['kernel_ms'], r['roofline']['frac'])"
if [ -n "${PMC:-}" ]; then
  B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
  run_pmc() { local name=$1; shift
    timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o r -- $B > "$OUT/pmc_$name.log" 2>&1; }
  run_pmc insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  run_pmc stall SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  find "$OUT" -name "*.csv" -size +8M -delete
fi
