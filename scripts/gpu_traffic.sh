#!/bin/bash
# Counters of the kernels behind the driver's bench line (the headline, its riders, the compress legs): FETCH_SIZE, WRITE_SIZE
# and the vector / scalar instruction counts per launch, SEPARATE rocprofv3 --pmc passes of the same commands
# (MI355X_MICROARCH.md: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2; never together with trace domains).
# scripts/collect_traffic.py turns the CSVs into profiles/pmc_traffic_r06.json, which bench.py replays for the same workload
# AND the same kernel sources only.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-traffic}
mkdir -p "$OUT"
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-riders"
run() { # name, bench args...   (ONLY="name name": just these, and the result is merged into the committed record)
  local name=$1; shift
  if [ -n "${ONLY:-}" ] && ! echo " $ONLY " | grep -q " $name "; then return; fi
  for ctr in FETCH_SIZE WRITE_SIZE INSTS; do
    local pmc=$ctr; [ $ctr = INSTS ] && pmc="SQ_INSTS_VALU SQ_INSTS_SALU"
    timeout 400 rocprofv3 --pmc $pmc --output-format csv -d "$OUT/${name}_$ctr" -o r -- $B "$@" > "$OUT/${name}_$ctr.log" 2>&1
    echo "$name $ctr rc=$?" | tee -a "$OUT/rc.txt"
  done
}
run lz4                                               # lz4_decompress_window_kernel + lz4_compress_wide_kernel (the extras leg)
run snappy --algo snappy                              # ... and Snappy's pair (BASELINE.json configs[2]: a round trip)
run deflate --algo deflate --no-extras --mib-per-gpu 1024 --unique-mib 32
run cascaded --algo cascaded --dataset example_float_columns --mib-per-gpu 1024 --unique-mib 32
run ans --algo ans --dataset silesia_style --mib-per-gpu 1024 --unique-mib 32
run bitcomp --algo bitcomp --dataset float_columns --mib-per-gpu 1024 --unique-mib 32
run lz4_mortgage --no-extras --dataset mortgage_col0_like --mib-per-gpu 1024 --unique-mib 64
run lz4_mortgage_default --no-extras --dataset mortgage_col0_like --producer fast --mib-per-gpu 1024 --unique-mib 64
run lz4_int32 --no-extras --dataset int32 --producer fast --mib-per-gpu 1024 --unique-mib 32
run lz4_mortgage_5120 --no-extras --dataset mortgage_col0_like --mib-per-gpu 320 --unique-mib 32
run lz4_16384 --no-extras --mib-per-gpu 1024
run lz4_4096 --no-extras --mib-per-gpu 256
run lz4_256 --no-extras --mib-per-gpu 16 --unique-mib 16
if [ "${LINES:-0}" = 1 ]; then # the other codecs' own lines at their default sizes
  run cascaded_line --algo cascaded --no-extras
  run bitcomp_line --algo bitcomp --no-extras
  run ans_line --algo ans --no-extras
  run deflate_line --algo deflate --no-extras
fi
find "$OUT" -name "*.csv" -size +16M -delete
python scripts/collect_traffic.py "$OUT" > "$OUT/pmc_traffic_r06.json" || exit 1
if [ -n "${ONLY:-}" ]; then # the other codecs' records stay as committed (their kernel sources have not changed: bench.py checks the digest)
  python - "$OUT/pmc_traffic_r06.json" <<'PY'
import json, sys
new = json.load(open(sys.argv[1])); old = json.load(open("profiles/pmc_traffic_r06.json"))
key = lambda r: (r["algo"], r["kind"], r["dataset"], r["chunks_per_gpu"], r.get("producer"))
fresh = {key(r) for r in new}
json.dump([r for r in old if key(r) not in fresh] + new, open(sys.argv[1], "w"), indent=1)
PY
fi
python -c "
import json; r=json.load(open('$OUT/pmc_traffic_r06.json'))
for x in r: print(x['algo'], x['kind'], x['dataset'], x['chunks_per_gpu'], 'traffic x', round(x['hbm_bytes_per_launch']/x['algorithmic_bytes'],2), 'valu', x.get('valu_wave_insts'))"
