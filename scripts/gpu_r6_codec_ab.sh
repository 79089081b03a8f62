#!/bin/bash
# Same-box A/B of one codec's bench line: the shipped library ("new") against builds under nvcomp_amd/lib/alt
# (scripts/build_api_variant.sh), alternating, three rounds, at 1 GiB and at the codec's default size; the codec's GPU tests first.
# usage: [VARIANTS="tag ..."] [TESTS=0] [DATASETS="name ..."] gpu_r6_codec_ab.sh <algo> <out tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
ALGO=$1; OUT=gpurun_out/${2:-r6ab_$1}; mkdir -p "$OUT"
[ "${TESTS:-1}" = 1 ] && timeout 900 python -m pytest tests/test_$ALGO.py tests/test_golden_decode.py tests/test_programs.py -m gpu -q -x 2>&1 | tail -3 | tee "$OUT/pytest.log"
line() { python -c "
import json,sys; r=json.loads(sys.stdin.read()); e=r.get('extras',{}); print('$1', 'dec', r['value'], 'frac', r['roofline']['frac'], 'ms_per_step', r['ms_per_step'], 'comp', e.get('gpu_compress_GBps'))"; }
for rep in 1 2 3; do
  for which in new ${VARIANTS:-}; do
    if [ $which != new ]; then export NVCOMP_AMD_LIB=$PWD/nvcomp_amd/lib/alt/libnvcomp_$which.so; else unset NVCOMP_AMD_LIB; fi
    for ds in ${DATASETS:-default}; do
      dsarg=""; [ $ds != default ] && dsarg="--dataset $ds"
      python bench.py --algo $ALGO $dsarg --mib-per-gpu 1024 --unique-mib 32 --no-cpu-baseline 2>/dev/null | tee -a "$OUT/lines_${which}.jsonl" | line "$which $ds 1GiB"
      python bench.py --algo $ALGO $dsarg --no-cpu-baseline 2>/dev/null | tee -a "$OUT/lines_${which}.jsonl" | line "$which $ds default"
    done
  done
done
