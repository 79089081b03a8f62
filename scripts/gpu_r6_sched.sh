#!/bin/bash
# Round 6: the other codecs' kernels compiled with -amdgpu-sched-strategy=max-ilp (nvcomp_amd/lib/ilpall/libnvcomp.so: the
# library's Makefile with that flag for every file) against the shipped build: bench.py lines at 1 GiB, alternating.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r6sched}; mkdir -p "$OUT"
for rep in 1 2; do
for a in cascaded ans bitcomp deflate; do
  ds=""; [ $a = cascaded ] && ds="--dataset example_float_columns"
  for lib in default ilpall; do
    if [ $lib = default ]; then unset NVCOMP_AMD_LIB; else export NVCOMP_AMD_LIB=$PWD/nvcomp_amd/lib/ilpall/libnvcomp.so; fi
    timeout 300 python bench.py --algo $a $ds --mib-per-gpu 1024 --unique-mib 32 --no-cpu-baseline --no-riders --steps 10 --warmup 2 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); e=r.get('extras',{}); print(json.dumps({'algo':'$a','lib':'$lib','rep':$rep,'dec':r['value'],'comp':e.get('gpu_compress_GBps'),'ratio':e.get('gpu_compress_ratio')}))" | tee -a "$OUT/sched.jsonl"
  done
done
done
unset NVCOMP_AMD_LIB
