#!/bin/bash
# hardware counters of the LZ4 compressor (1 GiB silesia-style mix): which unit is busy -- SQ issue, TA/TCP, LDS?
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$REPO"
OUT=$REPO/gpurun_out/${1:-cpmc}
mkdir -p "$OUT"
export TMPDIR=/tmp
B="python $REPO/scripts/bench_roundtrip.py --algo ${ALGO:-lz4} --dataset ${DATASET:-silesia_style} --unique-mib 32 --mib 1024 --iters 2"
cd /tmp
rocprofv3 --list-avail > "$OUT/avail.txt" 2>&1 || rocprofv3 -L > "$OUT/avail.txt" 2>&1
pass() { local name=$1; shift
  timeout 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o r -- $B > "$OUT/pmc_$name.log" 2>&1; echo "pmc $name rc=$?" >> "$OUT/rc.txt"; }
pass insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass active SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS
pass ta TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass ta2 TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum
pass tcp TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass tcp2 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]; res = {}
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        if "compress" not in k or "decompress" in k: continue
        for c, v in cs.items(): res.setdefault(k, {})[c] = sum(v) / len(v)
json.dump(res, open(out + "/summary.json", "w"), indent=1); print(json.dumps(res, indent=1))
PY
cat "$OUT/rc.txt"
