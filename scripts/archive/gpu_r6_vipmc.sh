#!/bin/bash
# Round 6: table d of scripts/microbench/valu_issue (do the 2-cycle and the 4-cycle operations overlap?) and what the SQ
# counters say about kernels whose cycles per instruction are known (calibration of SQ_ACTIVE_INST_VALU / SQ_INST_CYCLES...)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; export TMPDIR=/tmp
TAG=${1:-r6d}; OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"; : > "$OUT/rc.txt"
timeout 300 scripts/microbench/valu_issue 2048 d > "$OUT/valu_issue.jsonl" 2> "$OUT/valu_issue.err"; echo "valu_issue rc=$?" >> "$OUT/rc.txt"
python - "$OUT" <<'PY'
import json, sys, os
for l in open(os.path.join(sys.argv[1], "valu_issue.jsonl")):
    x = json.loads(l)
    if "op" in x: print("%-46s w=%d cyc/simd=%.2f wave=%.2f wall=%.2f ghz=%.2f" % (x["op"], x["waves_per_simd_launched"], x["cycles_per_instr_per_simd_median"], x["cycles_per_instr_one_wave_median"], x["cycles_per_instr_per_simd_from_wall"], x["memtime_ghz_median"]))
PY
cd /tmp
pass() { local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o r -- $REPO/scripts/microbench/valu_issue 2048 d 4 > "$OUT/pmc_$name.log" 2>&1; echo "pmc $name rc=$?" >> "$OUT/rc.txt"; }
pass insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass busy SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES
pass valu SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]; res = {}
for f in sorted(glob.glob(out + "/pmc_*/*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        d = res.setdefault(k, {}).setdefault(row["Counter_Name"], [])
        d.append(float(row["Counter_Value"]))
summ = {k: {c: v[-1] for c, v in cs.items()} for k, cs in res.items()}   # the second (warm) dispatch
json.dump(summ, open(out + "/pmc_summary.json", "w"), indent=1)
for k, cs in summ.items(): print(k, {c: round(v) for c, v in cs.items()})
PY
find "$OUT" -name "*.csv" -size +8M -delete
cat "$OUT/rc.txt"
