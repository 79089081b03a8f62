#!/bin/bash
# Calibration of FETCH_SIZE for this kernel's access patterns (MI355X_MICROARCH.md "HBM": calibrate on a known byte count).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-calib}
mkdir -p "$OUT"
pmc() { # name, env..., -- bench args
  local name=$1; shift
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$name" -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify "$@" > "$OUT/$name.log" 2>&1
  python - "$OUT/$name/r_counter_collection.csv" "$name" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    if "decompress_window_kernel" in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
for k, v in agg.items():
    print(sys.argv[2], k, "per launch KB:", v / len(n[k]), "launches", len(n[k]))
PY
}
pmc noise --dataset noise --producer fast
pmc zeros --dataset zeros --producer fast
NVCOMP_AMD_LIB=$PWD/nvcomp_amd/lib/alt/libnvcomp_farabl.so pmc farabl
pmc mix
