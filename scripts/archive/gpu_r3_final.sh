#!/bin/bash
# Round-3 evidence run: smoke + full GPU test tier, HBM traffic passes FIRST (scripts/gpu_traffic.sh -> profiles/archive/pmc_traffic_r03.json,
# which the bench lines below then replay), the driver's bench line and the other codecs' lines, rocprofv3 kernel stats of
# the same commands, PMC instruction / stall passes (LZ4, Snappy, mortgage-like), N sweep, data-class sweep, HLIF.
# usage: gpu_r3_final.sh <tag> [notests]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
REPO=$PWD
TAG=${1:-r3final}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"; : > "$OUT/rc.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" >> "$OUT/rc.txt"
if [ "${2:-}" != "notests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/rc.txt"
fi
bash scripts/gpu_traffic.sh "$TAG/traffic" > "$OUT/traffic.log" 2>&1; echo "traffic rc=$?" >> "$OUT/rc.txt"
[ -s "$OUT/traffic/pmc_traffic_r03.json" ] && cp "$OUT/traffic/pmc_traffic_r03.json" profiles/archive/pmc_traffic_r03.json
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
timeout 600 python scripts/ab_decode.py --libs nvcomp_amd/lib/libnvcomp.so --cases mix16m,mix64m,mix128m,mix256m,mix512m,mix1g,mix,snappy64m,snappy256m,snappy_mix --out "$OUT/nsweep.jsonl" > /dev/null 2> "$OUT/nsweep.err"
timeout 600 python scripts/ab_decode.py --libs nvcomp_amd/lib/libnvcomp.so --cases mortgage,mortgage_hc,mortgage5k,int32,zeros,noise,text,snappy_mortgage,snappy_int32,snappy_zeros,snappy_noise --out "$OUT/classes.jsonl" > /dev/null 2> "$OUT/classes.err"
timeout 400 python scripts/ab_compress.py --libs nvcomp_amd/lib/libnvcomp.so --out "$OUT/compress.jsonl" > /dev/null 2> "$OUT/compress.err"
for spec in "lz4 silesia_style" "lz4 mortgage_col0_like" "snappy silesia_style" "cascaded example_float_columns" "ans silesia_style"; do
  set -- $spec
  timeout 200 python scripts/bench_roundtrip.py --algo $1 --dataset $2 --unique-mib 32 --mib 1024 >> "$OUT/roundtrip.jsonl" 2>> "$OUT/roundtrip.err"
done
timeout 200 python scripts/bench_roundtrip.py --algo bitcomp --dataset float_columns --opts 0,4 --unique-mib 32 --mib 1024 >> "$OUT/roundtrip.jsonl" 2>> "$OUT/roundtrip.err"
bash scripts/gpu_hlif_crc.sh "$TAG/hlif" > "$OUT/hlif.txt" 2>&1
find "$OUT" -name "*.csv" -size +6M -delete; find "$OUT" -name "*.db" -delete
cat "$OUT/rc.txt"; [ -f "$OUT/pytest_gpu.log" ] && tail -2 "$OUT/pytest_gpu.log"
python - "$OUT" <<'PY'
import json,sys,os
o=sys.argv[1]
r=json.load(open(os.path.join(o,"bench_lz4.json")))
print("LZ4", r["value"], r["roofline"], "riders", {k:(v.get("value") if isinstance(v,dict) else v) for k,v in r["extras"].items() if isinstance(v,(dict,int,float))})
for name in ("nsweep.jsonl", "classes.jsonl"):
    for l in open(os.path.join(o,name)):
        x=json.loads(l); print(x["case"], x.get("chunks"), x.get("GBps"), x.get("ok"))
PY
