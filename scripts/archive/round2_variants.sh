#!/bin/bash
# Local build of the A/B libraries prepared (not yet measured) at the end of round 1; run before
# `gpurun -- 'bash scripts/gpu_round2_ab.sh <tag>'`. See DESIGN.md section 6 and profiles/archive/r01_occupancy_variants.json.
set -e
cd "$(dirname "$0")/.."
rm -f nvcomp_amd/lib/alt/*.so
bash scripts/build_variants.sh \
  h3072o6 "-DNVCOMP_LZM_HASH_ENTRIES=3072 -DNVCOMP_LZM_WAVES_PER_SIMD=6" \
  h11o8 "-DNVCOMP_LZM_HASH_BITS=11 -DNVCOMP_LZM_WAVES_PER_SIMD=8" \
  lzw1472o7 "-DNVCOMP_LZW_OUTWIN=1472 -DNVCOMP_LZW_BATCHMAX=736 -DNVCOMP_LZW_KEEP=544 -DNVCOMP_LZW_WAVES_PER_SIMD=7" \
  lzw1728o7 "-DNVCOMP_LZW_OUTWIN=1728 -DNVCOMP_LZW_BATCHMAX=864 -DNVCOMP_LZW_KEEP=640 -DNVCOMP_LZW_WAVES_PER_SIMD=7"
