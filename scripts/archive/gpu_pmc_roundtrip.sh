#!/bin/bash
# PMC counters of the compress / decompress kernels of one codec's round trip (scripts/bench_roundtrip.py)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-pr}; shift
mkdir -p "$OUT"
for pass in "insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  set -- $pass "$@"
  name=$1; shift
  ctrs=""; while [ $# -gt 0 ] && [[ "$1" == SQ_* ]]; do ctrs="$ctrs $1"; shift; done
  timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/$name" -o r -- python scripts/bench_roundtrip.py --iters 2 --unique-mib 32 --mib 1024 "$@" > "$OUT/$name.log" 2>&1
  python - "$OUT/$name/r_counter_collection.csv" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-60:]
    if "compress" in k:
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, v in agg.items():
    print(k, {c: round(x / len(n[k]) / 1e6, 1) for c, x in sorted(v.items())}, "launches", len(n[k]))
PY
done
tail -2 "$OUT/insts.log"
