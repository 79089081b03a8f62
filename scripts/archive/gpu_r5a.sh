#!/bin/bash
# Round 5, GPU session 1: the 256-position-step LZ compressors (common/lz_match_wide.hip.h) on hardware -- the compress
# tests, then every build under nvcomp_amd/lib/cab/ (scripts/build_comp_variants.sh) on the same uploaded batches.
# usage: gpu_r5a.sh <tag> [cases]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-r5a}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"; : > "$OUT/rc.txt"
timeout 600 python -m pytest tests/test_lz4_encode.py tests/test_snappy.py -m gpu -q -x --timeout 300 > "$OUT/pytest_enc.log" 2>&1; echo "pytest enc rc=$?" >> "$OUT/rc.txt"
tail -3 "$OUT/pytest_enc.log"
CASES=${2:-mix,snappy_mix,text,int32,mortgage,noise}
timeout 1500 python scripts/ab_compress.py --libs nvcomp_amd/lib/libnvcomp.so $(ls nvcomp_amd/lib/cab/libnvcomp_*.so 2>/dev/null) \
  --cases $CASES --steps 5 --prof --out "$OUT/ab_comp.jsonl" > /dev/null 2> "$OUT/ab_comp.err"; echo "ab comp rc=$?" >> "$OUT/rc.txt"
python - "$OUT" <<'PY'
import json, sys, os
for l in open(os.path.join(sys.argv[1], "ab_comp.jsonl")):
    x = json.loads(l); print(x["case"], x["lib"], x.get("GBps"), x.get("ratio"), x.get("ok"), x.get("error", ""), x.get("phase_share", ""))
PY
tail -5 "$OUT/ab_comp.err"
if [ "${BENCH:-1}" = 1 ]; then
  timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?" >> "$OUT/rc.txt"
  python -c "
import json,sys
r=json.load(open('$OUT/bench.json')); e=r['extras']
print('lz4', r['value'], 'comp', e['gpu_compress_GBps'], e['gpu_compress_ratio'], 'snappy', e['snappy']['value'], 'casc', e['cascaded']['value'], 'n16k', e.get('lz4_16384',{}).get('value'), 'n4k', e.get('lz4_4096',{}).get('value'))"
fi
cat "$OUT/rc.txt"
