#!/bin/bash
# benchmark_hlif with and without checksum verification (LZ4 / Snappy managers, 1 GiB of the mix), three runs each
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-hlifcrc}
mkdir -p "$OUT"
python - <<'PY'
import sys
sys.path.insert(0, ".")
from nvcomp_amd import datasets
datasets.silesia_style(1024 << 20, 1).tofile("/tmp/mix.bin")
PY
for round in 1 2 3; do for fmt in lz4 snappy; do for pol in 0 4; do
  echo "== $fmt checksum policy $pol" | tee -a "$OUT/hlif.log"
  timeout 300 benchmarks/bin/benchmark_hlif $fmt -f /tmp/mix.bin -n 10 --checksum $pol 2>&1 | grep "throughput\|ERROR" | tee -a "$OUT/hlif.log"
done; done; done
timeout 300 examples/bin/standard_crc_checksum 2>&1 | tail -2 | tee -a "$OUT/hlif.log"
