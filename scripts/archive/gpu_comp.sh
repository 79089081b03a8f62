#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-comp}
mkdir -p "$OUT"
timeout 300 python -m pytest tests/test_lz4_encode.py tests/test_snappy.py -m gpu -x -q --timeout 300 2>&1 | tail -2
timeout 900 python scripts/ab_compress.py --prof --cases ${CASES:-mix,snappy_mix,int32,text} --out "$OUT/abc.jsonl" 2> "$OUT/abc.err" | python -c "
import sys, json, collections
rows = collections.OrderedDict()
for l in sys.stdin:
    r = json.loads(l)
    rows.setdefault(r['case'], []).append(r)
    if 'phase_share' in r: print(r['case'], r['lib'], r['phase_share'])
for c, rs in rows.items():
    print(c, ' '.join('%s=%s@%s%s' % (r['lib'], r.get('GBps', 'ERR'), r.get('ratio'), '' if r.get('ok', False) else '!') for r in rs))
"
tail -3 "$OUT/abc.err"
