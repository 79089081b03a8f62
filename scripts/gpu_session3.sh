#!/bin/bash
# Parity + A/B of decoder variants + counters for the default variant.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-s3}
mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/rc.txt"
timeout 400 python bench.py --steps 10 --warmup 2 > "$OUT/bench_lz4.json" 2> "$OUT/bench_lz4.err"; echo "bench lz4 rc=$?" >> "$OUT/rc.txt"
NVCOMP_AMD_LZ4_DECODE=direct timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_lz4_direct.json" 2> "$OUT/bench_lz4_direct.err"
timeout 300 python bench.py --steps 10 --warmup 2 --unchecked --no-cpu-baseline --no-extras > "$OUT/bench_lz4_unchecked.json" 2> "$OUT/bench_lz4_unchecked.err"
timeout 400 python bench.py --algo snappy --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/bench_snappy.json" 2> "$OUT/bench_snappy.err"; echo "bench snappy rc=$?" >> "$OUT/rc.txt"
timeout 600 python scripts/bench_sweep.py --out "$OUT/sweep.jsonl" --mib 512 --unique-mib 32 --steps 5 > "$OUT/sweep.log" 2>&1; echo "sweep rc=$?" >> "$OUT/rc.txt"
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
run_pmc() { local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o r -- $B > "$OUT/pmc_$name.log" 2>&1; echo "pmc $name rc=$?" >> "$OUT/rc.txt"; }
run_pmc insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run_pmc stall SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
run_pmc lds SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o r -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/trace.log" 2>&1
find "$OUT" -name "*.csv" -size +8M -delete
cat "$OUT/rc.txt"; tail -3 "$OUT/pytest_gpu.log"; cat "$OUT/bench_lz4.json"
