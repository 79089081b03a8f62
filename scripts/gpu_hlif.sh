#!/bin/bash
# benchmark_hlif (one manager, one buffer) for every format on synthetic files
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-hlif}
mkdir -p "$OUT"
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, ".")
from nvcomp_amd import datasets
datasets.silesia_style(256 << 20, 1).tofile("/tmp/mix.bin")
datasets.int32_column(256 << 20, 1).tofile("/tmp/col.bin")
PY
for spec in "lz4 /tmp/mix.bin" "snappy /tmp/mix.bin" "ans /tmp/mix.bin" "cascaded /tmp/col.bin -t int" "bitcomp /tmp/col.bin -t int"; do
  set -- $spec
  fmt=$1; file=$2; shift 2
  echo "== $fmt"
  timeout 300 benchmarks/bin/benchmark_hlif $fmt -f $file -n 5 "$@" 2>&1 | grep -v "^---" | tee -a "$OUT/hlif.log"
done
