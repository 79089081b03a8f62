#!/bin/bash
# round 2: the reference's published shape (doc/Benchmarks.md:88-95: Mortgage col 0, int64, 329 MB = 5 021 chunks, LZ4 ratio 38.89;
# A100: 95.87 compress / 320.70 decompress GB/s) on nvcomp_amd.datasets.mortgage_col0_like, plus the N sweep of the mix
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-r2d}
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_lz4_decode.py tests/test_snappy.py tests/test_fuzz_decode.py -m gpu -q -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -2 "$OUT/pytest.log"
B="python bench.py --no-cpu-baseline --no-extras"
timeout 300 $B --dataset mortgage_col0_like --producer fast --unique-mib 314 --mib-per-gpu 314 --steps 20 --warmup 3 > "$OUT/mortgage_lz4_cpu.json" 2> "$OUT/mortgage_lz4_cpu.err"
timeout 300 $B --algo snappy --dataset mortgage_col0_like --unique-mib 314 --mib-per-gpu 314 --steps 20 --warmup 3 > "$OUT/mortgage_snappy_cpu.json" 2>> "$OUT/mortgage_lz4_cpu.err"
for algo in lz4 snappy; do
  timeout 300 python scripts/bench_roundtrip.py --algo $algo --dataset mortgage_col0_like --unique-mib 314 --mib 314 --iters 10 >> "$OUT/mortgage_roundtrip.jsonl" 2>> "$OUT/mortgage_lz4_cpu.err"
done
timeout 300 python scripts/bench_roundtrip.py --algo lz4 --dataset int32 --unique-mib 32 --mib 1024 >> "$OUT/mortgage_roundtrip.jsonl" 2>> "$OUT/mortgage_lz4_cpu.err"
python - "$OUT" <<'PY'
import json,sys,os
o=sys.argv[1]
for f in ("mortgage_lz4_cpu.json","mortgage_snappy_cpu.json"):
    try:
        r=json.load(open(os.path.join(o,f))); print(f, r["value"], "GB/s ratio", r["config"]["ratio"], "chunks", r["config"]["chunks_per_gpu"], "frac", r["roofline"]["frac"])
    except Exception as e: print(f,"ERR",e)
for l in open(os.path.join(o,"mortgage_roundtrip.jsonl")):
    r=json.loads(l); print(r["algo"], r["dataset"], "chunks", r["chunks"], "ratio", r["ratio"], "comp", r["compress_GBps"], "decomp", r["decompress_GBps"])
PY
for mib in 256 64 16; do
  timeout 300 $B --steps 20 --warmup 3 --mib-per-gpu $mib > "$OUT/mix_$mib.json" 2>> "$OUT/mortgage_lz4_cpu.err"
  python -c "
import json; r=json.load(open('$OUT/mix_$mib.json')); print('mix mib $mib chunks', r['config']['chunks_per_gpu'], r['value'], 'GB/s')"
done
