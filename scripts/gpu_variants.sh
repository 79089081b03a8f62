#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-var}
mkdir -p "$OUT"
for lib in nvcomp_amd/lib/alt/libnvcomp_*.so; do
  tag=$(basename $lib .so)
  NVCOMP_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-verify > "$OUT/$tag.json" 2> "$OUT/$tag.err"
  NVCOMP_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-verify --dataset text --producer fast --mib-per-gpu 512 --unique-mib 32 > "$OUT/${tag}_text.json" 2>> "$OUT/$tag.err"
  NVCOMP_AMD_LIB=$PWD/$lib timeout 300 python bench.py --algo snappy --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-verify > "$OUT/${tag}_snappy.json" 2>> "$OUT/$tag.err"
  python - "$OUT/$tag.json" "$OUT/${tag}_text.json" "$OUT/${tag}_snappy.json" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        r=json.load(open(f)); print(f, r['value'], r['roofline']['kernel_ms'])
    except Exception as e: print(f,'ERR',e)
PY
done
if [ -n "${PMC_LIB:-}" ]; then
  B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify"
  run_pmc() { local name=$1; shift
    NVCOMP_AMD_LIB=$PWD/$PMC_LIB timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o r -- $B > "$OUT/pmc_$name.log" 2>&1; }
  run_pmc insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  run_pmc stall SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  run_pmc fetch FETCH_SIZE
  run_pmc lds SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
  find "$OUT" -name "*.csv" -size +8M -delete
fi
