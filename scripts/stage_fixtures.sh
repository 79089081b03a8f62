#!/bin/bash
# The reference's two input files for the harness programs -> benchmarks/data/ (git-ignored: reference DATA travels to the GPU
# box with the snapshot, it is never committed). In this container they come from /root/reference/benchmarks; without the
# reference, ExampleFloatData.csv is rebuilt from the committed golden LZ4 vectors (tests/golden/manifest.json holds its md5).
set -e
cd "$(dirname "$0")/.."
mkdir -p benchmarks/data
if [ -d /root/reference/benchmarks ]; then
  cp /root/reference/benchmarks/ExampleTable.txt /root/reference/benchmarks/ExampleFloatData.csv benchmarks/data/
else
  python - <<'PY'
import json, hashlib, numpy as np
from oracle import oracle_py as o
o.build()
m = json.load(open("tests/golden/manifest.json"))["files"]["ExampleFloatData.csv"]
parts = [o.lz4_decompress(np.fromfile("tests/golden/" + c["streams"]["lz4_hc12"]["file"], dtype=np.uint8), c["bytes"])[1][: c["bytes"]] for c in m["chunks"]]
data = np.concatenate(parts)
assert hashlib.md5(data.tobytes()).hexdigest() == m["md5"]
data.tofile("benchmarks/data/ExampleFloatData.csv")
PY
fi
# the three float32 columns of ExampleFloatData.csv (benchmarks/text_to_binary.py, the reference's command line)
for c in 0 1 2; do cp tests/golden/ExampleFloatData_col${c}_float.bin benchmarks/data/col${c}.float32.bin; done
ls -la benchmarks/data
