#!/bin/bash
# Round 5: counters of the LZ compressors (instructions, L1 lookups, fabric traffic: separate --pmc passes) and the round
# trip of the GPU-compressed mix at 16 384 chunks beside the HC-compressed one (scripts/ab_decode.py case mix1g).
# usage: gpu_compress_counters.sh <tag>
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
export TMPDIR=/tmp
TAG=${1:-r5c}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; : > "$OUT/rc.txt"
for algo in lz4 snappy; do
  timeout 300 python scripts/bench_roundtrip.py --algo $algo --dataset silesia_style --unique-mib 64 --mib 1024 2>> "$OUT/err.log" | tee -a "$OUT/roundtrip.jsonl" | cut -c1-400
done
timeout 300 python scripts/ab_decode.py --libs nvcomp_amd/lib/libnvcomp.so --cases mix1g --steps 5 --warmup 2 --out "$OUT/ab_dec.jsonl" > /dev/null 2>> "$OUT/err.log"
cat "$OUT/ab_dec.jsonl" | cut -c1-300
for algo in lz4 snappy; do
  ALGO=$algo bash scripts/gpu_comp_pmc.sh $TAG/pmc_$algo > "$OUT/pmc_$algo.log" 2>&1
  B="python $REPO/scripts/bench_roundtrip.py --algo $algo --dataset silesia_style --unique-mib 64 --mib 1024 --iters 2"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 200 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc_${algo}_$ctr" -o r -- $B > "$OUT/pmc_${algo}_$ctr.log" 2>&1); echo "$algo $ctr rc=$?" >> "$OUT/rc.txt"
  done
done
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]; res = {}
for algo in ("lz4", "snappy"):
    r = {}
    try:
        r.update(list(json.load(open(f"{out}/pmc_{algo}/summary.json")).values())[0])
    except Exception as e:
        r["insts_error"] = str(e)
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"{out}/pmc_{algo}_{ctr}/**/*counter_collection.csv", recursive=True):
            rows = [x for x in csv.DictReader(open(f)) if "_compress_" in x["Kernel_Name"]]
            n = len({x["Dispatch_Id"] for x in rows})
            r[ctr + "_KB_per_launch"] = sum(float(x["Counter_Value"]) for x in rows) / max(1, n)
    if "FETCH_SIZE_KB_per_launch" in r:
        r["fabric_bytes_per_launch"] = (2 * r["FETCH_SIZE_KB_per_launch"] + r.get("WRITE_SIZE_KB_per_launch", 0)) * 1024
    res[algo] = r
json.dump(res, open(out + "/compress_counters.json", "w"), indent=1); print(json.dumps(res, indent=1))
PY
find "$OUT" -name "*.csv" -size +8M -delete
cat "$OUT/rc.txt"
