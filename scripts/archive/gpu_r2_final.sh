#!/bin/bash
# Round-2 evidence run (lean): full GPU test tier + smoke, the driver's bench line and the other codecs' lines, rocprofv3
# kernel stats of the same command, PMC passes (instructions, stalls, FETCH/WRITE: separate passes), N sweep, round trips.
# usage: gpu_r2_final.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
REPO=$PWD
OUT=$PWD/gpurun_out/${1:-r2final}
mkdir -p "$OUT"; : > "$OUT/rc.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" >> "$OUT/rc.txt"
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/rc.txt"
timeout 600 python bench.py > "$OUT/bench_lz4.json" 2> "$OUT/bench_lz4.err"; echo "bench lz4 rc=$?" >> "$OUT/rc.txt"
for a in snappy cascaded bitcomp ans deflate; do
  timeout 400 python bench.py --algo $a > "$OUT/bench_$a.json" 2> "$OUT/bench_$a.err"; echo "bench $a rc=$?" >> "$OUT/rc.txt"
done
B="python $REPO/bench.py --no-cpu-baseline --no-extras"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o r -- $B --steps 5 --warmup 1 > "$OUT/trace.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_snappy" -o r -- $B --algo snappy --steps 5 --warmup 1 > "$OUT/trace_snappy.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_deflate" -o r -- $B --algo deflate --steps 5 --warmup 1 > "$OUT/trace_deflate.log" 2>&1
run_pmc() { local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o r -- $B --steps 2 --warmup 1 > "$OUT/pmc_$name.log" 2>&1; echo "pmc $name rc=$?" >> "$OUT/rc.txt"; }
run_pmc insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run_pmc stall SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
run_pmc_s() { local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_snappy_$name" -o r -- $B --algo snappy --steps 2 --warmup 1 > "$OUT/pmc_snappy_$name.log" 2>&1; echo "pmc snappy $name rc=$?" >> "$OUT/rc.txt"; }
run_pmc_s fetch FETCH_SIZE
run_pmc_s write WRITE_SIZE
run_pmc_d() { local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_deflate_$name" -o r -- $B --algo deflate --steps 2 --warmup 1 > "$OUT/pmc_deflate_$name.log" 2>&1; echo "pmc deflate $name rc=$?" >> "$OUT/rc.txt"; }
run_pmc_d insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
run_pmc_d fetch FETCH_SIZE
run_pmc_d write WRITE_SIZE
cd "$REPO"
timeout 300 $B --algo deflate --mib-per-gpu 1024 --unique-mib 64 --steps 10 --warmup 2 2>> "$OUT/nsweep.err" >> "$OUT/nsweep.jsonl"
for algo in lz4 snappy; do for mib in 16 256 1024 4096; do
  u=$(( mib < 64 ? mib : 64 ))
  timeout 400 $B --algo $algo --mib-per-gpu $mib --unique-mib $u --steps 10 --warmup 2 2>> "$OUT/nsweep.err" >> "$OUT/nsweep.jsonl"
done; done
for spec in "lz4 silesia_style" "lz4 text" "lz4 int32" "lz4 mortgage_col0_like" "snappy silesia_style" "snappy int32" "cascaded int32" "cascaded example_float_columns" "ans silesia_style" "deflate silesia_style" "deflate int32"; do
  set -- $spec
  timeout 200 python scripts/bench_roundtrip.py --algo $1 --dataset $2 --unique-mib 32 --mib 1024 >> "$OUT/roundtrip.jsonl" 2>> "$OUT/roundtrip.err"
done
timeout 200 python scripts/bench_roundtrip.py --algo bitcomp --dataset float_columns --opts 0,4 --unique-mib 32 --mib 1024 >> "$OUT/roundtrip.jsonl" 2>> "$OUT/roundtrip.err"
timeout 300 $B --dataset mortgage_col0_like --producer fast --unique-mib 314 --mib-per-gpu 314 --steps 20 --warmup 3 > "$OUT/mortgage_lz4.json" 2>> "$OUT/nsweep.err"
find "$OUT" -name "*.csv" -size +6M -delete; find "$OUT" -name "*.db" -delete
cat "$OUT/rc.txt"; tail -2 "$OUT/pytest_gpu.log"; python - "$OUT" <<'PY'
import json,sys,os
o=sys.argv[1]
r=json.load(open(os.path.join(o,"bench_lz4.json")))
print("LZ4", r["value"], r["roofline"]["frac"], "snappy rider", r["extras"]["snappy"]["value"], "compress", r["extras"]["gpu_compress_GBps"], r["extras"]["gpu_compress_ratio"])
for l in open(os.path.join(o,"nsweep.jsonl")):
    x=json.loads(l); print(x["metric"], x["config"]["chunks_per_gpu"], x["value"])
for l in open(os.path.join(o,"roundtrip.jsonl")):
    x=json.loads(l); print(x["algo"], x["dataset"], x["ratio"], x["compress_GBps"], x["decompress_GBps"])
PY
