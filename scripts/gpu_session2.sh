#!/bin/bash
# Parity tests + instruction/stall/cache counters of the decode kernel.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-s2}
mkdir -p "$OUT"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" >> "$OUT/rc.txt"
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/rc.txt"
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras ${BENCH_ARGS:-}"
run_pmc() { # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o r -- $B > "$OUT/pmc_$name.log" 2>&1
  echo "pmc $name rc=$?" >> "$OUT/rc.txt"
}
run_pmc insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run_pmc stall SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run_pmc l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run_pmc l1 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
run_pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o r -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/trace.log" 2>&1
find "$OUT" -name "*.csv" -size +8M -delete
cat "$OUT/rc.txt"; tail -3 "$OUT/pytest_gpu.log"
