#!/bin/bash
# bench every A/B build under nvcomp_amd/lib/alt/ (scripts/build_variants.sh); output is not verified (ablations)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-var}
mkdir -p "$OUT"
for lib in nvcomp_amd/lib/libnvcomp.so nvcomp_amd/lib/alt/libnvcomp_*.so; do
  tag=$(basename $lib .so)
  NVCOMP_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-verify > "$OUT/$tag.json" 2> "$OUT/$tag.err"
  NVCOMP_AMD_LIB=$PWD/$lib timeout 300 python bench.py --algo snappy --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-verify > "$OUT/${tag}_snappy.json" 2>> "$OUT/$tag.err"
  python - "$OUT/$tag.json" "$OUT/${tag}_snappy.json" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        r=json.load(open(f)); print(f, r['value'], r['roofline']['kernel_ms'])
    except Exception as e: print(f,'ERR',e)
PY
done
