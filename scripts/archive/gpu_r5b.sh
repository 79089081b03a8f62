#!/bin/bash
# Round 5, GPU session 2: compress A/B (scripts/build_comp_variants.sh builds under nvcomp_amd/lib/cab/) and a decode A/B of
# the given decoder builds under nvcomp_amd/lib/alt/ against the default one (scripts/ab_decode.py, one process).
# usage: gpu_r5b.sh <tag> [compress cases] [decode libs (tags under lib/alt)] [decode cases]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-r5b}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"; : > "$OUT/rc.txt"
CASES=${2:-mix,snappy_mix,text,int32,mortgage,noise}
if [ "$CASES" != none ]; then
timeout 1500 python scripts/ab_compress.py --libs nvcomp_amd/lib/libnvcomp.so $(ls nvcomp_amd/lib/cab/libnvcomp_*.so 2>/dev/null) \
  --cases $CASES --steps 5 --prof --out "$OUT/ab_comp.jsonl" > /dev/null 2> "$OUT/ab_comp.err"; echo "ab comp rc=$?" >> "$OUT/rc.txt"
python - "$OUT" <<'PY'
import json, sys, os
for l in open(os.path.join(sys.argv[1], "ab_comp.jsonl")):
    x = json.loads(l); print(x["case"], x["lib"], x.get("GBps"), x.get("ratio"), x.get("ok"), x.get("error", ""), x.get("phase_share", ""))
PY
tail -3 "$OUT/ab_comp.err"
fi
DLIBS=${3:-}
if [ -n "$DLIBS" ]; then
  L="nvcomp_amd/lib/libnvcomp.so"; for t in $DLIBS; do L="$L nvcomp_amd/lib/alt/libnvcomp_$t.so"; done
  timeout 1200 python scripts/ab_decode.py --libs $L --cases ${4:-mix,snappy_mix,mix1g,mix256m,mortgage} --steps 5 --warmup 2 \
    --out "$OUT/ab_dec.jsonl" > /dev/null 2> "$OUT/ab_dec.err"; echo "ab dec rc=$?" >> "$OUT/rc.txt"
  python - "$OUT" <<'PY'
import json, sys, os
for l in open(os.path.join(sys.argv[1], "ab_dec.jsonl")):
    x = json.loads(l); print(x["case"], x.get("chunks"), x["lib"], x.get("GBps"), x.get("ok"), x.get("error", ""))
PY
  tail -3 "$OUT/ab_dec.err"
fi
cat "$OUT/rc.txt"
