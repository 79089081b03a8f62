#!/bin/bash
# The evidence run of a round (ONE script since round 6; the one-off sessions of rounds 1-5 are under scripts/archive/):
# smoke + full GPU test tier, HBM traffic passes FIRST (scripts/gpu_traffic.sh -> profiles/pmc_traffic_r<NN>.json, which the
# bench lines below then replay), the driver's bench line and the other codecs' lines, rocprofv3 kernel stats of the same
# commands, PMC instruction / stall passes (LZ4, Snappy, mortgage-like), N sweep, data-class sweep, round trips, HLIF,
# phase clocks, the harness programs. scripts/collect_evidence.py <tag> r<NN>_final copies the judged summaries to profiles/.
# usage: gpu_evidence.sh <tag> [notests]      (SKIP="harness pmc ..." leaves parts out)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
REPO=$PWD
TAG=${1:-evidence}
PMC_RECORD=$(python -c 'import bench; print(bench.PMC_RECORD)')
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"; : > "$OUT/rc.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" >> "$OUT/rc.txt"
if [ "${2:-}" != "notests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/rc.txt"
fi
bash scripts/gpu_traffic.sh "$TAG/traffic" > "$OUT/traffic.log" 2>&1; echo "traffic rc=$?" >> "$OUT/rc.txt"
[ -s "$OUT/traffic/$PMC_RECORD" ] && cp "$OUT/traffic/$PMC_RECORD" profiles/$PMC_RECORD
timeout 600 python bench.py > "$OUT/bench_lz4.json" 2> "$OUT/bench_lz4.err"; echo "bench lz4 rc=$?" >> "$OUT/rc.txt"
for a in snappy cascaded bitcomp ans deflate; do
  timeout 400 python bench.py --algo $a --no-riders > "$OUT/bench_$a.json" 2> "$OUT/bench_$a.err"; echo "bench $a rc=$?" >> "$OUT/rc.txt"
done
B="python $REPO/bench.py --no-cpu-baseline --no-extras"
cd /tmp
for spec in "trace:" "trace_snappy:--algo snappy" "trace_deflate:--algo deflate" "trace_cascaded:--algo cascaded" "trace_bitcomp:--algo bitcomp" "trace_ans:--algo ans" "trace_compress:--extras-compress-only"; do
  name=${spec%%:*}; args=${spec#*:}
  if [ "$name" = trace_compress ]; then
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -o r -- python $REPO/bench.py --no-cpu-baseline --no-riders --steps 3 --warmup 1 > "$OUT/$name.log" 2>&1
  else
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -o r -- $B $args --steps 5 --warmup 1 > "$OUT/$name.log" 2>&1
  fi
done
cd "$REPO"
ALGOS="lz4 snappy" bash scripts/gpu_pmc.sh "$TAG/pmc" > "$OUT/pmc.log" 2>&1
EXTRA='--dataset mortgage_col0_like --mib-per-gpu 1024 --unique-mib 64' bash scripts/gpu_pmc.sh "$TAG/pmc_mortgage" > "$OUT/pmc_mortgage.log" 2>&1
timeout 300 $B --algo deflate --mib-per-gpu 1024 --unique-mib 64 --steps 10 --warmup 2 2>> "$OUT/nsweep.err" >> "$OUT/deflate_1g.json"
# batch-size sweep of the mix, one process (scripts/ab_decode.py: 256 .. 65 536 chunks)
timeout 900 python scripts/ab_decode.py --libs nvcomp_amd/lib/libnvcomp.so nvcomp_amd/lib/alt/libnvcomp_team.so nvcomp_amd/lib/alt/libnvcomp_pair.so nvcomp_amd/lib/alt/libnvcomp_chase.so --cases mix1m,mix4m,mix16m,mix32m,mix64m,mix128m,mix256m,mix512m,mix1g,snappy16m,snappy32m,snappy64m,snappy256m --out "$OUT/nsweep.jsonl" > /dev/null 2> "$OUT/nsweep.err"
timeout 600 python scripts/ab_decode.py --libs nvcomp_amd/lib/libnvcomp.so --cases mix,snappy_mix --out "$OUT/nsweep.jsonl" > /dev/null 2>> "$OUT/nsweep.err"
# the request rate of the memory system inside / outside the Infinity Cache (scripts/probes/gather_calib.hip: footprint 128 MiB / 2 GiB)
for mib in 64 128 512 2048; do timeout 120 scripts/probes/gather_calib 16 $mib >> "$OUT/calib_mall.jsonl" 2>> "$OUT/nsweep.err"; done
bash scripts/kernel_resources.sh -- lz4_api snappy_api cascaded_api deflate_api bitcomp_api ans_api pack_api 2>/dev/null | sort > "$OUT/kernel_resources.txt"
timeout 600 python scripts/ab_decode.py --libs nvcomp_amd/lib/libnvcomp.so --cases mortgage,mortgage_hc,mortgage5k,int32,zeros,noise,text,snappy_mortgage,snappy_int32,snappy_zeros,snappy_noise --out "$OUT/classes.jsonl" > /dev/null 2> "$OUT/classes.err"
timeout 400 python scripts/ab_compress.py --libs nvcomp_amd/lib/libnvcomp.so --out "$OUT/compress.jsonl" > /dev/null 2> "$OUT/compress.err"
for spec in "lz4 silesia_style" "lz4 mortgage_col0_like" "snappy silesia_style" "cascaded example_float_columns" "ans silesia_style"; do
  set -- $spec
  timeout 200 python scripts/bench_roundtrip.py --algo $1 --dataset $2 --unique-mib 32 --mib 1024 >> "$OUT/roundtrip.jsonl" 2>> "$OUT/roundtrip.err"
done
timeout 200 python scripts/bench_roundtrip.py --algo bitcomp --dataset float_columns --opts 0,4 --unique-mib 32 --mib 1024 >> "$OUT/roundtrip.jsonl" 2>> "$OUT/roundtrip.err"
bash scripts/gpu_hlif_crc.sh "$TAG/hlif" > "$OUT/hlif.txt" 2>&1
# counters of the LZ compressors (instructions, L1 lookups, fabric traffic) + the GPU-compressed mix through the decoder
bash scripts/gpu_compress_counters.sh "$TAG/comp" > "$OUT/comp_counters.log" 2>&1
# phase clocks: the compressors' (-DNVCOMP_LZM_PROF), the Cascaded decoder's (-DNVCOMP_CASC_PROF), the window decoders' (-DNVCOMP_LZW_PROF)
if ls nvcomp_amd/lib/cab/libnvcomp_wprof.so > /dev/null 2>&1; then
  timeout 300 python scripts/ab_compress.py --libs nvcomp_amd/lib/cab/libnvcomp_wprof.so --cases mix,text,int32,noise --steps 3 --prof --out "$OUT/compress_phases.jsonl" > /dev/null 2>> "$OUT/compress.err"
fi
if ls nvcomp_amd/lib/cab/libnvcomp_cascprof.so > /dev/null 2>&1; then
  for ds in example_float_columns int32; do
    NVCOMP_AMD_LIB=$PWD/nvcomp_amd/lib/cab/libnvcomp_cascprof.so timeout 200 python scripts/casc_prof.py $ds 1024 2>/dev/null | tail -1 >> "$OUT/cascaded_phases.jsonl"
    NVCOMP_AMD_LIB=$PWD/nvcomp_amd/lib/cab/libnvcomp_cascprof.so timeout 200 python scripts/casc_prof.py $ds 1024 --compress 2>/dev/null | tail -1 | sed 's/^{/{"with_compress_leg": true, /' >> "$OUT/cascaded_phases.jsonl"
  done
fi
LZWPROF=$(ls nvcomp_amd/lib/cab/libnvcomp_lzwprof.so nvcomp_amd/lib/alt/libnvcomp_prof.so 2>/dev/null | head -1)
if [ -n "$LZWPROF" ]; then
  for a in lz4 snappy; do
    NVCOMP_AMD_PROF=1 NVCOMP_AMD_LIB=$PWD/$LZWPROF timeout 300 python bench.py --algo $a --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep phase_share | sed "s/^/{\"algo\": \"$a\", \"prof\": /; s/$/}/" >> "$OUT/decode_phases.jsonl"
  done
fi
# Cascaded: one batch size per rocprofv3 summary, 1 GiB and 4 GiB of the reference's float columns
cd /tmp
for spec in "trace_cascaded_1g:--mib-per-gpu 1024 --unique-mib 32" "trace_cascaded_4g:"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -o r -- python $REPO/bench.py --algo cascaded --dataset example_float_columns --no-cpu-baseline --no-extras $args --steps 5 --warmup 1 > "$OUT/$name.log" 2>&1
done
cd "$REPO"
[ "${HARNESS:-1}" = 1 ] && bash scripts/gpu_harness.sh "$TAG/harness" > "$OUT/harness_run.log" 2>&1
find "$OUT" -name "*.csv" -size +6M -delete; find "$OUT" -name "*.db" -delete
cat "$OUT/rc.txt"; [ -f "$OUT/pytest_gpu.log" ] && tail -2 "$OUT/pytest_gpu.log"
python - "$OUT" <<'PY'
import json,sys,os
o=sys.argv[1]
r=json.load(open(os.path.join(o,"bench_lz4.json")))
print("LZ4", r["value"], r["roofline"], "riders", {k:(v.get("value") if isinstance(v,dict) else v) for k,v in r["extras"].items() if isinstance(v,(dict,int,float))})
for name in ("nsweep.jsonl", "classes.jsonl"):
    for l in open(os.path.join(o,name)):
        x=json.loads(l); print(x["case"], x.get("lib"), x.get("chunks"), x.get("GBps"), x.get("ok"))
PY
