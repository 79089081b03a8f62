#!/bin/bash
# hardware counters of the LZ4 compressor (1 GiB silesia-style mix): instructions per window, L1 lookups, busy cycles
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
OUT=$REPO/gpurun_out/${1:-cpmc}
mkdir -p "$OUT"
export TMPDIR=/tmp
B="python $REPO/scripts/bench_roundtrip.py --algo ${ALGO:-lz4} --dataset ${DATASET:-silesia_style} --unique-mib 32 --mib 1024 --iters 2"
cd /tmp
pass() { local name=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o r -- $B > "$OUT/pmc_$name.log" 2>&1; echo "pmc $name rc=$?" >> "$OUT/rc.txt"; }
pass insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]; res = {}
for f in glob.glob(out + "/pmc_*/*counter_collection.csv"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        if "_compress_" not in k: continue
        for c, v in cs.items(): res.setdefault(k.split("(")[0][-40:], {})[c] = sum(v) / len(v)
json.dump(res, open(out + "/summary.json", "w"), indent=1); print(json.dumps(res, indent=1))
PY
cat "$OUT/rc.txt"
