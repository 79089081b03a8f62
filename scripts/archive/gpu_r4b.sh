#!/bin/bash
# Round 4, GPU session 2: the workgroup-per-chunk LZ4 decoder (common/lz_team.hip.h) on hardware -- parity tests of the
# forced-team build, then the batch-size sweep against the one-wave and two-wave paths (scripts/ab_decode.py, one process).
# usage: gpu_r4b.sh <tag> [cases]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-r4b}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"; : > "$OUT/rc.txt"
ALT=nvcomp_amd/lib/alt
timeout 900 python -m pytest tests/test_lz4_decode.py tests/test_golden_decode.py tests/test_fuzz_decode.py tests/test_fuzz_corrupt.py -m gpu -q -x -k "team and not nappy" --timeout 600 > "$OUT/pytest_team.log" 2>&1; echo "pytest team rc=$?" >> "$OUT/rc.txt"
tail -3 "$OUT/pytest_team.log"
CASES=${2:-mix16m,mix64m,mix128m,mix256m,mix512m,mix1g,text,mortgage5k,int32,zeros,noise}
timeout 1200 python scripts/ab_decode.py --libs $ALT/libnvcomp_chase.so $ALT/libnvcomp_pair.so $ALT/libnvcomp_team.so \
  --cases $CASES --steps 5 --warmup 2 --out "$OUT/ab_team.jsonl" > /dev/null 2> "$OUT/ab_team.err"; echo "ab team rc=$?" >> "$OUT/rc.txt"
python - "$OUT" <<'PY'
import json, sys, os, collections
rows = collections.defaultdict(dict)
for l in open(os.path.join(sys.argv[1], "ab_team.jsonl")):
    x = json.loads(l); rows[(x["case"], x.get("chunks"))][x["lib"]] = (x.get("GBps"), x.get("ok"), x.get("error"))
for k, v in rows.items():
    print(k, v)
PY
cat "$OUT/rc.txt"
