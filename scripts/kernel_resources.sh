#!/bin/bash
# registers, spills, LDS and occupancy of every kernel of the given api/*.hip files (default: the LZ ones), one line each
# usage: kernel_resources.sh [extra hipcc flags] -- [lz4_api snappy_api ...]
cd "$(dirname "$0")/.."
FLAGS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do FLAGS+=("$1"); shift; done; [ "${1:-}" = "--" ] && shift
FILES=${*:-lz4_api snappy_api deflate_api}
for f in $FILES; do
  SCHED=(); case $f in lz4_api|snappy_api) SCHED=(-mllvm -amdgpu-sched-strategy=max-ilp);; esac # as nvcomp_amd/csrc/Makefile (LZ_SCHED)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Invcomp_amd/csrc -Wno-unused-function "${SCHED[@]}" "${FLAGS[@]}" \
    -Rpass-analysis=kernel-resource-usage -c nvcomp_amd/csrc/api/$f.hip -o /dev/null 2>&1 |
    python3 -c "
import re, sys
cur = {}
for l in sys.stdin:
    m = re.search(r'remark: +(Function Name|TotalSGPRs|VGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)', l)
    if not m: continue
    k, v = m.groups()
    if k == 'Function Name':
        cur = {'name': re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', v)[:44]}
    cur[k.split(' [')[0]] = v
    if k.startswith('LDS'):
        print('%-44s sgpr %3s vgpr %3s spill s%s v%s scratch %s occ %s lds %s' % (cur['name'], cur.get('TotalSGPRs'), cur.get('VGPRs'), cur.get('SGPRs Spill'), cur.get('VGPRs Spill'), cur.get('ScratchSize'), cur.get('Occupancy'), cur.get('LDS Size')))
" &
done; wait
