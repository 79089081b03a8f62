#!/bin/bash
# Round-end evidence run: full parity suite, headline bench lines, A/B variants, sweeps, rocprofv3 summaries.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-final}
mkdir -p "$OUT"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" >> "$OUT/rc.txt"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/rc.txt"
timeout 400 python bench.py > "$OUT/bench_lz4.json" 2> "$OUT/bench_lz4.err"; echo "bench lz4 rc=$?" >> "$OUT/rc.txt"
timeout 300 python bench.py --unchecked --no-cpu-baseline --no-extras > "$OUT/bench_lz4_unchecked.json" 2> "$OUT/bench_lz4_unchecked.err"
timeout 400 python bench.py --algo snappy > "$OUT/bench_snappy.json" 2> "$OUT/bench_snappy.err"; echo "bench snappy rc=$?" >> "$OUT/rc.txt"
timeout 600 python scripts/bench_sweep.py --out "$OUT/sweep.jsonl" --mib 512 --unique-mib 32 --steps 5 > "$OUT/sweep.log" 2>&1; echo "sweep rc=$?" >> "$OUT/rc.txt"
for ds in int32 float_columns float32 lowcard noise; do
  timeout 200 python scripts/bench_roundtrip.py --algo cascaded --dataset $ds --mib 1024 --unique-mib 32 >> "$OUT/roundtrip.jsonl" 2>> "$OUT/roundtrip.err"
done
for algo in lz4 snappy; do for ds in silesia_style text int32; do
  timeout 200 python scripts/bench_roundtrip.py --algo $algo --dataset $ds --unique-mib 32 --mib 1024 >> "$OUT/roundtrip.jsonl" 2>> "$OUT/roundtrip.err"
done; done
for ds in silesia_style text table float_csv int32 lowcard zeros noise; do
  timeout 200 python scripts/bench_roundtrip.py --algo ans --dataset $ds --unique-mib 32 --mib 1024 >> "$OUT/roundtrip.jsonl" 2>> "$OUT/roundtrip.err"
done
for spec in "int32 0,4" "int32 1,4" "float32 0,4" "float_columns 0,4" "int32 0,6" "int32 0,2" "silesia_style 0,1" "zeros 0,1"; do
  set -- $spec
  timeout 200 python scripts/bench_roundtrip.py --algo bitcomp --dataset $1 --opts $2 --unique-mib 32 --mib 1024 >> "$OUT/roundtrip.jsonl" 2>> "$OUT/roundtrip.err"
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_ans" -o r -- python scripts/bench_roundtrip.py --algo ans --dataset silesia_style --unique-mib 32 --mib 1024 > "$OUT/trace_ans.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_bitcomp" -o r -- python scripts/bench_roundtrip.py --algo bitcomp --dataset int32 --opts 0,4 --unique-mib 32 --mib 1024 > "$OUT/trace_bitcomp.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_cascaded" -o r -- python scripts/bench_roundtrip.py --algo cascaded --dataset int32 --unique-mib 32 --mib 1024 > "$OUT/trace_cascaded.log" 2>&1
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
run_pmc() { local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o r -- $B > "$OUT/pmc_$name.log" 2>&1; echo "pmc $name rc=$?" >> "$OUT/rc.txt"; }
run_pmc insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run_pmc stall SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o r -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/trace.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_snappy" -o r -- python bench.py --algo snappy --steps 5 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/trace_snappy.log" 2>&1
find "$OUT" -name "*.csv" -size +8M -delete
cat "$OUT/rc.txt"; tail -3 "$OUT/pytest_gpu.log"; cat "$OUT/bench_lz4.json"
for a in cascaded bitcomp ans; do
  timeout 400 python bench.py --algo $a > "$OUT/bench_$a.json" 2> "$OUT/bench_$a.err"; echo "bench $a rc=$?" >> "$OUT/rc.txt"
done
bash scripts/gpu_traffic.sh "${1:-final}" > "$OUT/traffic.log" 2>&1
tail -3 "$OUT/rc.txt"
