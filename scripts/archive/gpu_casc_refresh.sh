#!/bin/bash
# A change to the Cascaded kernels only, on hardware in one short call: its GPU tests, a same-box A/B of the 1 GiB float
# columns against nvcomp_amd/lib/cab/libnvcomp_cascbase.so (the build of the sources before the change), the counter passes of
# the Cascaded records alone (merged into pmc_traffic_r05.json: the other codecs' records stay, their sources have not
# changed), the codec's own 4 GiB line and the driver's line with its riders. usage: gpu_casc_refresh.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-cascr}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
timeout 200 python -m pytest tests/test_cascaded.py tests/test_cascaded_pins.py tests/test_golden_decode.py -m gpu -q -x 2>&1 | tail -2 | tee "$OUT/pytest.log"
B="python bench.py --algo cascaded --dataset example_float_columns --mib-per-gpu 1024 --unique-mib 32 --no-cpu-baseline"
for rep in 1 2; do
  for lib in nvcomp_amd/lib/cab/libnvcomp_cascbase.so nvcomp_amd/lib/libnvcomp.so; do
    [ -f $lib ] || continue
    NVCOMP_AMD_LIB=$PWD/$lib timeout 120 $B 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); e=r.get('extras',{}); print(json.dumps({'lib':'$lib'.split('_')[-1],'decompress':r['value'],'frac':r['roofline']['frac'],'compress':e.get('gpu_compress_GBps'),'ratio':e.get('gpu_compress_ratio')}))" | tee -a "$OUT/ab.jsonl"
  done
done
ONLY="cascaded" bash scripts/gpu_traffic.sh "$TAG/traffic" > "$OUT/traffic.log" 2>&1; echo "traffic rc=$?" | tee -a "$OUT/rc.txt"
[ -s "$OUT/traffic/pmc_traffic_r05.json" ] && cp "$OUT/traffic/pmc_traffic_r05.json" profiles/pmc_traffic_r05.json
timeout 200 python bench.py --algo cascaded --no-riders > "$OUT/bench_cascaded.json" 2> "$OUT/bench_cascaded.err"; echo "bench cascaded rc=$?" | tee -a "$OUT/rc.txt"
timeout 300 python bench.py > "$OUT/bench_lz4.json" 2> "$OUT/bench_lz4.err"; echo "bench lz4 rc=$?" | tee -a "$OUT/rc.txt"
python - "$OUT" <<'PY'
import json, sys, os
o = sys.argv[1]
c = json.load(open(os.path.join(o, "bench_cascaded.json"))); l = json.load(open(os.path.join(o, "bench_lz4.json")))
print("cascaded line", c["value"], c["roofline"]["frac"], c["extras"].get("gpu_compress_GBps"), (c["extras"].get("compress_roofline") or {}).get("frac"))
r = l["extras"]["cascaded"]; print("rider", r["value"], r["roofline"]["frac"], r["roofline"]["traffic"], r["compress"]["value"], r["compress"]["roofline"]["frac"], r["compress"]["roofline"]["traffic"])
print("lz4 line", l["value"], l["roofline"]["traffic"], l["extras"]["gpu_compress_GBps"], l["extras"]["compress_roofline"]["traffic"])
PY
