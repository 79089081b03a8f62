#!/bin/bash
# Round 6: the token index (common/lz_index.hip.h) on the card: its tests, the LZ decode tests, an A/B against builds under
# nvcomp_amd/lib/alt (noidx = -DNVCOMP_LZ_INDEX=0: the round-5 chase everywhere), optionally the bench line.
# usage: gpu_r6e.sh <tag> "<pytest files / -k ...>" "<alt lib tags>" "<ab cases>" [bench: 0|1]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-r6e}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"; : > "$OUT/rc.txt"
TESTS=${2:-tests/test_token_index.py tests/test_lz4_decode.py tests/test_headline_parity.py}
if [ "$TESTS" != none ]; then
  timeout 1500 python -m pytest $TESTS -x -q -m gpu > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/rc.txt"; tail -5 "$OUT/pytest.log"
fi
DLIBS=${3:-noidx}
if [ "$DLIBS" != none ]; then
  L="nvcomp_amd/lib/libnvcomp.so"; for t in $DLIBS; do L="$L nvcomp_amd/lib/alt/libnvcomp_$t.so"; done
  timeout 1200 python scripts/ab_decode.py --libs $L --cases ${4:-mix,mix1g,text,mortgage,noise} --steps 5 --warmup 2 \
    --out "$OUT/ab_dec.jsonl" > /dev/null 2> "$OUT/ab_dec.err"; echo "ab dec rc=$?" >> "$OUT/rc.txt"
  python - "$OUT" <<'PY'
import json, sys, os
for l in open(os.path.join(sys.argv[1], "ab_dec.jsonl")):
    x = json.loads(l); print(x["case"], x.get("chunks"), x["lib"], x.get("GBps"), x.get("ok"), x.get("error", ""))
PY
  tail -3 "$OUT/ab_dec.err"
fi
if [ "${5:-0}" = 1 ]; then
  timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?" >> "$OUT/rc.txt"
  tail -c 2500 "$OUT/bench.json"
fi
cat "$OUT/rc.txt"
