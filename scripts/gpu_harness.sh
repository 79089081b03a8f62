#!/bin/bash
# Round-4 harness evidence: the reference-shaped C++ callers on the reference's own inputs
# (SURVEY.md 8(d) configs 1, 2(i), 3 class A, 4) -> gpurun_out/<tag>/harness.log (copied to profiles/r04_harness.log).
#   benchmark_{lz4,snappy}_chunked  -f ExampleTable.txt ExampleFloatData.csv -x 1200 (1.07 GB)   benchmark_template_chunked.cuh:340-353,590-617
#   benchmark_cascaded_chunked -t int on the three float32 columns, -x 22000 (1.06 GB)          benchmark_cascaded_chunked.cu:35-36
#   benchmark_snappy_synth (4 000 chunks of gen_data(3), the reference's defaults)              benchmark_snappy_synth.cpp:161-193
#   benchmark_lz4_synth (zeros / noise, 64 KiB x 2^b, b = 0 .. 13)                              benchmark_lz4_synth.cpp:64-72
#   examples/lz4_cpu_compression on ExampleFloatData.csv (config 1: liblz4 HC on the host, ratio = BASELINE.md section 2)
# usage: gpu_harness.sh <tag>   (scripts/stage_fixtures.sh must have run in the container: benchmarks/data/)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-r4harness}
mkdir -p "$OUT"
LOG=$OUT/harness.log
: > "$LOG"
make -s -C benchmarks -j8 > /dev/null 2>&1; make -s -C examples -j4 > /dev/null 2>&1
D=benchmarks/data
B=benchmarks/bin
say() { echo "### $*" | tee -a "$LOG"; }
run() { say "$*"; timeout 600 "$@" >> "$LOG" 2>&1; echo "rc=$?" >> "$LOG"; }
say "host: $(nproc) hardware threads; $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2)"
run $B/benchmark_lz4_chunked -f $D/ExampleTable.txt $D/ExampleFloatData.csv -x 1200 -i 5 -w 2
run $B/benchmark_lz4_chunked -f $D/ExampleTable.txt $D/ExampleFloatData.csv -x 300 -i 5 -w 2
run $B/benchmark_lz4_chunked -f $D/ExampleTable.txt $D/ExampleFloatData.csv -i 5 -w 2
run $B/benchmark_snappy_chunked -f $D/ExampleTable.txt $D/ExampleFloatData.csv -x 1200 -i 5 -w 2
run $B/benchmark_cascaded_chunked -f $D/col0.float32.bin $D/col1.float32.bin $D/col2.float32.bin -t int -x 22000 -i 5 -w 2
run $B/benchmark_snappy_synth
run $B/benchmark_snappy_synth -m 255
run $B/benchmark_lz4_synth
run $B/benchmark_hlif lz4 -f $D/ExampleTable.txt
run examples/bin/lz4_cpu_compression -f $D/ExampleFloatData.csv
run examples/bin/lz4_cpu_compression -f $D/ExampleTable.txt
run examples/bin/lz4_cpu_decompression -f $D/ExampleTable.txt $D/ExampleFloatData.csv
# config 1 as a throughput: liblz4 on the host cores over the same two files duplicated (oracle/_ref shim, one C thread per core)
python - >> "$LOG" 2>&1 <<'PY'
import os, time, numpy as np
from oracle import oracle_py as o
from nvcomp_amd import datasets
o.build()
files = [np.fromfile("benchmarks/data/" + f, dtype=np.uint8) for f in ("ExampleTable.txt", "ExampleFloatData.csv")]
chunks = [c for f in files for c in datasets.split_chunks(f, 65536)] * 400
threads = len(os.sched_getaffinity(0))
raw = sum(c.size for c in chunks)
for codec, name in ((o.LZ4_ENC_HC, "LZ4_compress_HC(12)"), (o.LZ4_ENC, "LZ4_compress_default")):
    t, outs, errs = o.batch_run(codec, chunks, [o.lz4_bound(c.size) + 64 for c in chunks], threads=threads, use_ref=True)
    comp = [x.copy() for x in outs]
    csz = sum(c.size for c in comp)
    t2, _, e2 = o.batch_run(o.LZ4_DEC, comp, [c.size for c in chunks], threads=threads, use_ref=True, repeats=3)
    print(f"### config 1 (CPU plumbing, {threads} threads, {raw} B = the two fixture files x 400): {name} {raw / t / 1e9:.2f} GB/s, "
          f"ratio {raw / csz:.4f}; LZ4_decompress_safe {raw / t2 / 1e9:.2f} GB/s; errors {errs + e2}")
PY
cat "$LOG" | grep -E "^###|throughput|ratio|rc=|validated|GB/s" | head -120
