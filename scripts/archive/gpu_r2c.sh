#!/bin/bash
# round 2 validation pass: the whole GPU test tier, the default bench line, HLIF vs LLIF on the same file
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-r2c}
mkdir -p "$OUT"
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?"
python - "$OUT/bench_default.json" <<'PY'
import json,sys
r=json.load(open(sys.argv[1]))
print("lz4", r["value"], r["roofline"]["frac"], "| snappy", r["extras"]["snappy"]["value"], r["extras"]["snappy"]["roofline"]["frac"],
      "| comp", r["extras"]["gpu_compress_GBps"], r["extras"]["gpu_compress_ratio"], r["extras"]["compress_roofline"]["frac"],
      "| cpu", r["cpu_baseline"]["value"], r["cpu_baseline"].get("compress", {}).get("value"))
PY
python - <<'PY'
from nvcomp_amd import datasets
datasets.silesia_style(256 << 20, 0).tofile("/tmp/mix256.bin")
PY
make -C benchmarks -j8 > "$OUT/make_bench.log" 2>&1
for p in "benchmark_lz4_chunked" "benchmark_hlif lz4" "benchmark_snappy_chunked" "benchmark_hlif snappy"; do
  echo "== $p"; timeout 300 benchmarks/bin/$p -f /tmp/mix256.bin 2>&1 | grep -E "throughput|ratio" | tee -a "$OUT/hlif_vs_llif.log"
done
