#!/bin/bash
# full GPU validation: smoke, the whole -m gpu suite, the default bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-full}
mkdir -p "$OUT"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee "$OUT/rc.txt"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -x > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/rc.txt"
tail -5 "$OUT/pytest_gpu.log"
timeout 600 python bench.py > "$OUT/bench_lz4.json" 2> "$OUT/bench_lz4.err"; echo "bench rc=$?" | tee -a "$OUT/rc.txt"
python - "$OUT/bench_lz4.json" <<'PY'
import json,sys
r=json.load(open(sys.argv[1]))
print("LZ4", r["value"], "GB/s frac", r["roofline"]["frac"], "kernel_ms", r["roofline"]["kernel_ms"], "cpu", r.get("cpu_baseline",{}).get("value"))
e=r.get("extras",{})
print("compress", e.get("gpu_compress_GBps"), e.get("gpu_compress_ratio"))
for k in ("snappy","deflate","cascaded","lz4_mortgage_like"):
    v=e.get(k,{})
    print(k, v.get("value"), v.get("roofline",{}).get("frac"), v.get("error"), (v.get("cpu_baseline") or {}).get("value"))
PY
tail -3 "$OUT/bench_lz4.err"
