#!/bin/bash
# Round 6, GPU session 1 (VERDICT r5 "Next round" item 1): (a) scripts/microbench/valu_issue -- cycles per wave64 instruction
# per SIMD; (b) the round-5 window decoder with -DNVCOMP_LZW_FAR_ABLATE at 28 and 16 waves per CU beside the shipped build
# (scripts/ab_decode.py, one process); (c) the default bench line of this box.
# usage: gpu_r6a.sh <tag> [decode libs (tags under lib/alt)] [decode cases]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-r6a}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"; : > "$OUT/rc.txt"
if [ -x scripts/microbench/valu_issue ]; then
  timeout 300 scripts/microbench/valu_issue 2048 ${VI_TABLE:-} > "$OUT/valu_issue.jsonl" 2> "$OUT/valu_issue.err"; echo "valu_issue rc=$?" >> "$OUT/rc.txt"
  python - "$OUT" <<'PY'
import json, sys, os
for l in open(os.path.join(sys.argv[1], "valu_issue.jsonl")):
    x = json.loads(l)
    if "op" in x: print("%-46s w=%d seen=%s cyc/simd=%.2f wave=%.2f wall=%.2f ghz=%.2f" % (x["op"], x["waves_per_simd_launched"], x["waves_per_simd_seen_min_med_max"], x["cycles_per_instr_per_simd_median"], x["cycles_per_instr_one_wave_median"], x["cycles_per_instr_per_simd_from_wall"], x["memtime_ghz_median"]))
    else: print(x)
PY
fi
DLIBS=${2:-farab farab_w4 w4}
if [ "$DLIBS" != none ]; then
  L="nvcomp_amd/lib/libnvcomp.so"; for t in $DLIBS; do L="$L nvcomp_amd/lib/alt/libnvcomp_$t.so"; done
  timeout 1200 python scripts/ab_decode.py --libs $L --cases ${3:-mix,snappy_mix,mix1g} --steps 5 --warmup 2 \
    --out "$OUT/ab_dec.jsonl" > /dev/null 2> "$OUT/ab_dec.err"; echo "ab dec rc=$?" >> "$OUT/rc.txt"
  python - "$OUT" <<'PY'
import json, sys, os
for l in open(os.path.join(sys.argv[1], "ab_dec.jsonl")):
    x = json.loads(l); print(x["case"], x.get("chunks"), x["lib"], x.get("GBps"), x.get("ok"), x.get("error", ""))
PY
  tail -3 "$OUT/ab_dec.err"
fi
if [ "${BENCH:-1}" = 1 ]; then
  timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?" >> "$OUT/rc.txt"
  python - "$OUT" <<'PY'
import json, sys, os
x = json.loads(open(os.path.join(sys.argv[1], "bench.json")).read().strip().splitlines()[-1])
print("bench", x["value"], x["ms_per_step"], x["roofline"]["frac"])
PY
fi
cat "$OUT/rc.txt"
