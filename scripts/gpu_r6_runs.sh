#!/bin/bash
# Round 6: the run executor (lzw::execute_run_batch) against the build without it (lib/alt/libnvcomp_noruns.so:
# scripts/build_variants.sh noruns "-DNVCOMP_LZW_RUNS=0"), same process, same uploaded batches.
# usage: gpu_r6_runs.sh <tag> [cases]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-r6q}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
CASES=${2:-mortgage_hc,mortgage,int32,zeros,text,mix1g,mix,snappy_mortgage,snappy_int32,snappy_mix}
if [ "${TESTS:-1}" = 1 ]; then
  timeout 900 python -m pytest tests/test_lz4_decode.py tests/test_snappy.py tests/test_golden_decode.py -m gpu -q -x --timeout 600 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee "$OUT/rc.txt"
  tail -3 "$OUT/pytest.log"
fi
LIBS="nvcomp_amd/lib/libnvcomp.so"
for t in ${VARIANTS:-noruns}; do LIBS="$LIBS nvcomp_amd/lib/alt/libnvcomp_$t.so"; done
timeout 900 python scripts/ab_decode.py --libs $LIBS --cases $CASES --again --out "$OUT/ab_dec.jsonl" > /dev/null 2> "$OUT/ab_dec.err"; echo "ab dec rc=$?" | tee -a "$OUT/rc.txt"
python - "$OUT/ab_dec.jsonl" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    x=json.loads(l); print(x["case"], x.get("chunks"), x.get("lib"), x.get("GBps"), x.get("ok"), x.get("error",""))
PY
