#!/bin/bash
# a few bench.py lines, one per argument string in $LINES (separated by ';')
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-lines}
mkdir -p "$OUT"
IFS=';' read -ra SPECS <<< "${LINES}"
i=0
for spec in "${SPECS[@]}"; do
  i=$((i+1))
  timeout 400 python bench.py --no-cpu-baseline $spec > "$OUT/line_$i.json" 2> "$OUT/line_$i.err"
  python - "$OUT/line_$i.json" "$spec" <<'PY'
import json,sys
try:
    r=json.load(open(sys.argv[1])); e=r.get("extras",{})
    print(sys.argv[2], "->", r["value"], "GB/s  frac", r["roofline"]["frac"], " compress", e.get("gpu_compress_GBps"), e.get("gpu_compress_ratio"))
except Exception as ex: print(sys.argv[2], "ERR", ex)
PY
done
