#!/bin/bash
# One gpurun call: parity tests, headline bench, variant/dataset sweep, rocprofv3 summaries.
# Everything is wrapped in `timeout`; logs land in gpurun_out/ (merged back by gpurun).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-s1}
mkdir -p "$OUT"
{ rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | sort | uniq -c | head; nproc; free -g | head -2; ls /opt/conda/lib/liblz4.so* 2>&1 | head -2; } > "$OUT/box.txt" 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" >> "$OUT/rc.txt"
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/rc.txt"
timeout 400 python bench.py --steps 10 --warmup 2 > "$OUT/bench_lz4.json" 2> "$OUT/bench_lz4.err"; echo "bench lz4 rc=$?" >> "$OUT/rc.txt"
timeout 300 python bench.py --steps 10 --warmup 2 --unchecked --no-cpu-baseline --no-extras > "$OUT/bench_lz4_unchecked.json" 2> "$OUT/bench_lz4_unchecked.err"
NVCOMP_AMD_LZ4_DECODE=serial timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_lz4_serial.json" 2> "$OUT/bench_lz4_serial.err"
timeout 400 python bench.py --algo snappy --steps 10 --warmup 2 > "$OUT/bench_snappy.json" 2> "$OUT/bench_snappy.err"; echo "bench snappy rc=$?" >> "$OUT/rc.txt"
timeout 600 python scripts/bench_sweep.py --out "$OUT/sweep.jsonl" --mib 512 --unique-mib 32 --steps 5 > "$OUT/sweep.log" 2>&1; echo "sweep rc=$?" >> "$OUT/rc.txt"
# per-kernel durations (kernel trace + stats only; counters go in their own passes below)
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/prof_lz4" -o lz4 -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/prof_lz4.log" 2>&1; echo "rocprof rc=$?" >> "$OUT/rc.txt"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o lz4 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o lz4 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/pmc_write.log" 2>&1
find "$OUT" -name "*.csv" -size +20M -delete
ls -laR "$OUT" | head -60 > "$OUT/listing.txt"
cat "$OUT/rc.txt"; tail -3 "$OUT/pytest_gpu.log"; cat "$OUT/bench_lz4.json"
