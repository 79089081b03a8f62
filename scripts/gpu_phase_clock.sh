#!/bin/bash
# Round 6: phase clocks (-DNVCOMP_LZW_PROF builds under nvcomp_amd/lib/alt: scripts/build_variants.sh prof "-DNVCOMP_LZW_PROF" ...)
# usage: gpu_phase_clock.sh <tag> "<lib tags>" "<algos>" [bench args]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-r6p}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
LIBS=${2:-prof prof_noidx}; ALGOS=${3:-lz4}; if [ $# -ge 3 ]; then shift 3; else shift $#; fi
for t in $LIBS; do for a in $ALGOS; do
  NVCOMP_AMD_PROF=1 NVCOMP_AMD_LIB=$PWD/nvcomp_amd/lib/alt/libnvcomp_$t.so timeout 300 python bench.py --algo $a --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-riders "$@" > "$OUT/prof_${t}_$a.json" 2> "$OUT/prof_${t}_$a.err"
  echo "$t $a $(python -c "import json;print(json.load(open('$OUT/prof_${t}_$a.json'))['value'])")"; tail -1 "$OUT/prof_${t}_$a.err"
done; done
