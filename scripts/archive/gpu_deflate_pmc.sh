#!/bin/bash
# hardware counters of the DEFLATE decoder (1 GiB, zlib level 9 mix)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
OUT=$REPO/gpurun_out/${1:-dpmc}
mkdir -p "$OUT"
export TMPDIR=/tmp
B="python $REPO/bench.py --algo deflate --mib-per-gpu 1024 --unique-mib 32 --no-cpu-baseline --steps 2 --warmup 1"
cd /tmp
pass() { local name=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o r -- $B > "$OUT/pmc_$name.log" 2>&1; echo "pmc $name rc=$?" >> "$OUT/rc.txt"; }
pass insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass act GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]; res = {}
for f in glob.glob(out + "/pmc_*/*counter_collection.csv"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        if "deflate_decompress" not in k: continue
        for c, v in cs.items(): res[c] = sum(v) / len(v)
json.dump(res, open(out + "/summary.json", "w"), indent=1); print(json.dumps(res, indent=1))
PY
cat "$OUT/rc.txt"
