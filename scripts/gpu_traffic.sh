#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per launch of the own-format decode kernels and of Snappy's (separate --pmc passes;
# LZ4's come from gpu_final.sh's pmc_fetch / pmc_write legs)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-tr}
mkdir -p "$OUT"
for algo in bitcomp ans cascaded snappy; do for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/${algo}_$ctr" -o r -- python bench.py --algo $algo --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/${algo}_$ctr.log" 2>&1
  python - "$OUT/${algo}_$ctr/r_counter_collection.csv" "$algo" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    if "_decompress_" in r["Kernel_Name"] and "size_kernel" not in r["Kernel_Name"] and sys.argv[2] in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
import json, os
for k, v in agg.items():
    print(sys.argv[2], k, "KB per launch:", round(v / len(n[k]), 1), "launches", len(n[k]))
    json.dump({"KB_per_launch": v / len(n[k]), "launches": len(n[k])},
              open(os.path.join(os.path.dirname(os.path.dirname(sys.argv[1])), f"traffic_{sys.argv[2]}_{k}.json"), "w"))
PY
done; done
