#!/bin/bash
# round 3, call A: the LZ decode tests on the persistent-wave launch, then the first A/B of decode variants
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3a}
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_lz4_decode.py tests/test_snappy.py tests/test_golden_decode.py tests/test_fuzz_decode.py tests/test_fuzz_corrupt.py tests/test_abi.py -m gpu -x -q --timeout 300 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee "$OUT/rc.txt"
tail -3 "$OUT/pytest.log"
timeout 900 python scripts/ab_decode.py --again --cases ${CASES:-mix,snappy_mix,mix1g,mortgage,mortgage5k,zeros,noise,int32,text} --out "$OUT/ab.jsonl" 2> "$OUT/ab.err" | python -c "
import sys, json, collections
rows = collections.OrderedDict()
for l in sys.stdin:
    r = json.loads(l)
    rows.setdefault(r['case'], []).append(r)
for c, rs in rows.items():
    print(c, ' '.join('%s=%s%s' % (r['lib'], r.get('GBps', 'ERR'), '' if r.get('ok', False) else '!') for r in rs))
"
tail -3 "$OUT/ab.err"
if [ -f nvcomp_amd/lib/prof/libnvcomp_prof.so ]; then
  NVCOMP_AMD_PROF=1 NVCOMP_AMD_LIB=$PWD/nvcomp_amd/lib/prof/libnvcomp_prof.so timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --mib-per-gpu 1024 > "$OUT/prof_lz4.json" 2> "$OUT/prof_lz4.err"; tail -1 "$OUT/prof_lz4.err"
fi
