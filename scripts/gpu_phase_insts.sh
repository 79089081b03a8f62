#!/bin/bash
# Instruction counts per decoder phase: ablated kernels (a1 = stop after the chase, a2 = after the parse) vs the full one.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-pi}
mkdir -p "$OUT"
for v in a1 a2 window; do
  NVCOMP_AMD_LZ4_DECODE=$v timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/$v" -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify --unchecked > "$OUT/$v.log" 2>&1
  python - "$OUT/$v/r_counter_collection.csv" "$v" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); n = set()
for r in csv.DictReader(open(sys.argv[1])):
    if "lz4_decompress_window_kernel" in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
print(sys.argv[2], {k: round(v / len(n) / 1e6, 1) for k, v in sorted(agg.items())}, "launches", len(n))
PY
done
