#!/bin/bash
# Cascaded, round 6: the passes behind the first as resident, looping grids (api/cascaded_api.hip) against the build in
# nvcomp_amd/lib/cab/libnvcomp_cascold.so (grids over the batch): GPU tests, the 1 GiB and 4 GiB lines alternating,
# small batches, and a kernel trace of the new build. usage: gpu_r6_casc.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r6casc}; mkdir -p "$OUT"
[ "${TESTS:-1}" = 1 ] && timeout 900 python -m pytest tests/test_cascaded.py tests/test_cascaded_pins.py tests/test_golden_decode.py tests/test_programs.py -m gpu -q -x 2>&1 | tail -3 | tee "$OUT/pytest.log"
line() { python -c "
import json,sys; r=json.loads(sys.stdin.read()); e=r.get('extras',{}); print('$1', 'dec', r['value'], 'frac', r['roofline']['frac'], 'kernel_ms', r['roofline'].get('kernel_ms'), 'ms_per_step', r['ms_per_step'], 'comp', e.get('gpu_compress_GBps'))"; }
# VARIANTS: tags of builds under nvcomp_amd/lib/cab (scripts/build_casc_variant.sh), each timed in turn with the shipped one ("new")
for rep in 1 2 3; do
  for which in new ${VARIANTS:-cascold}; do
    if [ $which != new ]; then export NVCOMP_AMD_LIB=$PWD/nvcomp_amd/lib/cab/libnvcomp_$which.so; else unset NVCOMP_AMD_LIB; fi
    python bench.py --algo cascaded --mib-per-gpu 1024 --unique-mib 32 --no-cpu-baseline 2>/dev/null | tee -a "$OUT/lines_${which}.jsonl" | line "$which 1GiB"
    python bench.py --algo cascaded --no-cpu-baseline 2>/dev/null | tee -a "$OUT/lines_${which}.jsonl" | line "$which 4GiB"
    for mib in ${SMALL:-16 64 256}; do
      python bench.py --algo cascaded --mib-per-gpu $mib --unique-mib 16 --no-cpu-baseline 2>/dev/null | tee -a "$OUT/lines_${which}.jsonl" | line "$which ${mib}MiB"
    done
  done
done
unset NVCOMP_AMD_LIB
rocprofv3 --kernel-trace --stats -d "$OUT/trace_1g" -o r -- python bench.py --algo cascaded --mib-per-gpu 1024 --unique-mib 32 --no-cpu-baseline --steps 5 --warmup 1 > "$OUT/trace_1g.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/trace_1g/**/r_kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "cascaded_decompress" in r["Kernel_Name"]]
    for r in rows[:9]:
        print("trace", r["Kernel_Name"][:70].split("(")[0][-40:], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "ns grid", r.get("Grid_Size_X", r.get("Grid_Size")))
PY
