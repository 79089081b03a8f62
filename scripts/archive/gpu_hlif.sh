#!/bin/bash
# benchmark_hlif (one manager, one buffer): every checksum policy on the LZ4 / Snappy managers, then every format
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-hlif}
mkdir -p "$OUT"
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, ".")
from nvcomp_amd import datasets
datasets.silesia_style(1024 << 20, 1).tofile("/tmp/mix.bin")
datasets.int32_column(256 << 20, 1).tofile("/tmp/col.bin")
PY
for fmt in lz4 snappy; do for pol in 0 1 4; do
  echo "== $fmt checksum policy $pol" | tee -a "$OUT/hlif.log"
  timeout 300 benchmarks/bin/benchmark_hlif $fmt -f /tmp/mix.bin -n 5 --checksum $pol 2>&1 | grep "throughput\|ERROR\|ratio" | tee -a "$OUT/hlif.log"
done; done
for spec in "ans /tmp/mix.bin" "deflate /tmp/mix.bin" "cascaded /tmp/col.bin -t int" "bitcomp /tmp/col.bin -t int"; do
  set -- $spec
  fmt=$1; file=$2; shift 2
  echo "== $fmt" | tee -a "$OUT/hlif.log"
  timeout 300 benchmarks/bin/benchmark_hlif $fmt -f $file -n 5 "$@" 2>&1 | grep "throughput\|ERROR\|ratio" | tee -a "$OUT/hlif.log"
done
timeout 300 examples/bin/standard_crc_checksum 2>&1 | tail -2 | tee -a "$OUT/hlif.log"
timeout 600 python -m pytest tests/test_programs.py -m gpu -q --timeout 600 2>&1 | tail -3 | tee -a "$OUT/hlif.log"
