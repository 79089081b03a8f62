#!/bin/bash
# instruction and stall counters of the LZ4 (and Snappy) window decoder on the headline workload: separate --pmc passes
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-pmc}
mkdir -p "$OUT"
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras ${EXTRA:-}"
run_pmc() { local name=$1; shift; local algo=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_${algo}_$name" -o r -- $B --algo $algo > "$OUT/pmc_${algo}_$name.log" 2>&1; echo "pmc $algo $name rc=$?" >> "$OUT/rc.txt"; }
for algo in ${ALGOS:-lz4}; do
  run_pmc insts $algo SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  run_pmc stall $algo SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = {}
for path in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    algo = path.split("/pmc_")[1].split("_")[0]
    rows = [r for r in csv.DictReader(open(path)) if "decompress_window_kernel" in r["Kernel_Name"]]
    n = len({r["Dispatch_Id"] for r in rows}) or 1
    agg = collections.defaultdict(float)
    for r in rows: agg[r["Counter_Name"]] += float(r["Counter_Value"])
    out.setdefault(algo, {}).update({k: v / n for k, v in agg.items()})
json.dump(out, open(sys.argv[1] + "/pmc.json", "w"), indent=1)
for algo, c in out.items():
    print(algo, {k: f"{v/1e6:.0f}M" for k, v in sorted(c.items())})
PY
find "$OUT" -name "*.csv" -size +8M -delete
