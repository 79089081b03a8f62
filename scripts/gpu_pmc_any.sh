#!/bin/bash
# Instruction / stall counters of whatever kernels a bench.py command runs (separate --pmc passes, no trace domains).
# usage: gpu_pmc_any.sh <tag> <kernel substring> <bench.py arguments ...>
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
export TMPDIR=/tmp
TAG=$1; KERN=$2; shift 2
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
B="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras $*"
cd /tmp
pass() { local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o r -- $B > "$OUT/pmc_$name.log" 2>&1; echo "pmc $name rc=$?" >> "$OUT/rc.txt"; }
pass insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES
pass busy GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
pass mem TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY
python - "$OUT" "$KERN" <<'PY'
import csv, glob, json, sys, collections
out, kern = sys.argv[1], sys.argv[2]; res = {}
for f in glob.glob(out + "/pmc_*/*counter_collection.csv"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for row in csv.DictReader(open(f)):
        if kern not in row["Kernel_Name"]: continue
        acc[row["Kernel_Name"].split("(")[0][-48:]][row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Kernel_Name"].split("(")[0][-48:]].add(row["Dispatch_Id"])
    for k, cs in acc.items():
        d = res.setdefault(k, {}); d["dispatches"] = len(n[k])
        for c, v in cs.items(): d[c] = v / len(n[k])
json.dump(res, open(out + "/summary.json", "w"), indent=1); print(json.dumps(res, indent=1))
PY
find "$OUT" -name "*.csv" -size +8M -delete
cat "$OUT/rc.txt"
