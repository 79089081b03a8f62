#!/bin/bash
# round 2: LZ decode with the gather executor -- parity subset, LZ4 + Snappy bench lines, kernel stats, instruction counters.
# usage: gpu_r2b.sh <tag> [notest] [nopmc]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-r2b}
mkdir -p "$OUT"
if [ "${2:-}" != "notest" ]; then
  timeout 900 python -m pytest tests/test_lz4_decode.py tests/test_golden_decode.py tests/test_snappy.py tests/test_fuzz_decode.py tests/test_fuzz_corrupt.py -m gpu -q --timeout 600 -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
fi
REPO=$PWD
B="python $REPO/bench.py --no-cpu-baseline --no-extras --lz-index-min-batch 1000000000"
for algo in ${ALGOS:-lz4 snappy}; do
  for mib in ${MIBS:-4096 1024 256}; do
    timeout 300 $B --algo $algo --steps 10 --warmup 2 --mib-per-gpu $mib > "$OUT/${algo}_${mib}.json" 2> "$OUT/${algo}_${mib}.err"
    python -c "
import json; r=json.load(open('$OUT/${algo}_${mib}.json')); print('$algo mib $mib', r['value'], 'GB/s', r['roofline']['kernel_ms'], 'ms', r['roofline']['frac'])"
  done
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o r -- $B --steps 5 --warmup 1 > "$OUT/prof.log" 2>&1
cp "$OUT/prof/"*kernel_stats.csv "$OUT/kernel_stats_lz4.csv" 2>/dev/null || find "$OUT/prof" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_lz4.csv" \;
head -4 "$OUT/kernel_stats_lz4.csv" | cut -c1-160
if [ "${3:-}" != "nopmc" ]; then
  run_pmc() { local name=$1; shift
    timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o r -- $B --steps 2 --warmup 1 > "$OUT/pmc_$name.log" 2>&1; echo "pmc $name rc=$?"; }
  run_pmc insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  run_pmc stall SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  python - "$OUT" <<'PY'
import csv, sys, collections, glob, json, os
out = sys.argv[1]
res = {}
for name in ("insts", "stall"):
    for p in glob.glob(os.path.join(out, "pmc_" + name, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(p)))
        kern = [r for r in rows if "decompress_window_kernel" in r["Kernel_Name"]]
        launches = len({r["Dispatch_Id"] for r in kern}) or 1
        agg = collections.defaultdict(float)
        for r in kern:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
        res.update({k: v / launches for k, v in agg.items()})
json.dump(res, open(os.path.join(out, "pmc.json"), "w"), indent=1)
print(json.dumps(res))
PY
fi
cd "$OUT"; find . -name "*.csv" -size +2M -delete; find . -name "*.db" -delete
