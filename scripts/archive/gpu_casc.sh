#!/bin/bash
# Cascaded on hardware: GPU tests, the float / int columns at 1 GiB, the 4 GiB line, the phase clock of a prof build.
# usage: gpu_casc.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-casc}; mkdir -p "$OUT"
timeout 300 python -m pytest tests/test_cascaded.py tests/test_cascaded_pins.py tests/test_golden_decode.py -m gpu -q -x 2>&1 | tail -2
line() { python -c "
import json,sys; r=json.loads(sys.stdin.read()); e=r.get('extras',{}); print('$1', r['config']['dataset'], 'dec', r['value'], 'frac', r['roofline']['frac'], 'ratio', r['config']['ratio'], 'comp', e.get('gpu_compress_GBps'), (e.get('compress_roofline') or {}).get('frac'))"; }
for ds in example_float_columns float_columns int32 mortgage_col0_like; do
  python bench.py --algo cascaded --dataset $ds --mib-per-gpu 1024 --unique-mib 32 --no-cpu-baseline 2>/dev/null | tee -a "$OUT/lines.jsonl" | line 1GiB
done
python bench.py --algo cascaded --no-cpu-baseline 2>/dev/null | tee -a "$OUT/lines.jsonl" | line 4GiB
if [ -f nvcomp_amd/lib/cab/libnvcomp_cascprof.so ]; then
  NVCOMP_AMD_LIB=$PWD/nvcomp_amd/lib/cab/libnvcomp_cascprof.so python scripts/casc_prof.py example_float_columns 1024 2>&1 | tail -1 | cut -c1-500 | tee "$OUT/phases.json"
fi
