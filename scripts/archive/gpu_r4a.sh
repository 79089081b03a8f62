#!/bin/bash
# Round 4, GPU session 1 (VERDICT r3 item 1a + item 7):
#  * far-match fetch against the number of OPEN chunks: the shipped decoder with 1 / 2 / 4 / 7 persistent workgroups per CU
#    (4 / 8 / 16 / 28 open chunks per CU = 1 024 ... 7 168 on the card), timing (scripts/ab_decode.py, one process) and one
#    FETCH_SIZE pass each on the headline batch;
#  * the price of LDS-served far matches: a build whose output window holds the WHOLE chunk (67 KiB of LDS per wave, two
#    waves per CU, every match copied LDS -> LDS), timed at 512 ... 65 536 chunks;
#  * FETCH_SIZE calibration for scattered 16-byte loads (scripts/probes/gather_calib.hip).
# usage: gpu_r4a.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
REPO=$PWD
TAG=${1:-r4a}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"; : > "$OUT/rc.txt"
ALT=nvcomp_amd/lib/alt
timeout 900 python scripts/ab_decode.py --libs nvcomp_amd/lib/libnvcomp.so $ALT/libnvcomp_wg4.so $ALT/libnvcomp_wg2.so $ALT/libnvcomp_wg1.so \
  --cases mix,mix1g,snappy_mix --steps 5 --warmup 2 --out "$OUT/ab_open_chunks.jsonl" > /dev/null 2> "$OUT/ab_open_chunks.err"; echo "ab open rc=$?" >> "$OUT/rc.txt"
timeout 900 python scripts/ab_decode.py --libs nvcomp_amd/lib/libnvcomp.so $ALT/libnvcomp_chase.so $ALT/libnvcomp_ldsall.so \
  --cases mix16m,mix64m,mix256m,mix1g,mix,text,mortgage --steps 3 --warmup 1 --out "$OUT/ab_ldsall.jsonl" > /dev/null 2> "$OUT/ab_ldsall.err"; echo "ab ldsall rc=$?" >> "$OUT/rc.txt"
cd /tmp
for v in default wg4 wg2 wg1 ldsall; do
  lib=$REPO/$ALT/libnvcomp_$v.so; [ $v = default ] && lib=$REPO/nvcomp_amd/lib/libnvcomp.so
  for ctr in FETCH_SIZE WRITE_SIZE; do
    [ $ctr = WRITE_SIZE ] && [ $v != default ] && [ $v != ldsall ] && continue
    NVCOMP_AMD_LIB=$lib timeout 400 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc_${v}_$ctr" -o r -- \
      python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-riders > "$OUT/pmc_${v}_$ctr.log" 2>&1
    echo "pmc $v $ctr rc=$?" >> "$OUT/rc.txt"
  done
done
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/calib_FETCH" -o r -- $REPO/scripts/probes/gather_calib > "$OUT/calib.jsonl" 2> "$OUT/calib.err"; echo "calib rc=$?" >> "$OUT/rc.txt"
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d "$OUT/calib_RDREQ" -o r -- $REPO/scripts/probes/gather_calib 8 > "$OUT/calib_rdreq.jsonl" 2> "$OUT/calib_rdreq.err"; echo "calib rdreq rc=$?" >> "$OUT/rc.txt"
cd "$REPO"
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*.csv" -size +8M -delete
python - "$OUT" <<'PY'
import csv, glob, json, os, sys, collections
o = sys.argv[1]
res = {}
for d in sorted(glob.glob(os.path.join(o, "pmc_*_*_SIZE")) + glob.glob(os.path.join(o, "calib_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: [0.0, set()])
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:60], r["Counter_Name"])
            agg[k][0] += float(r["Counter_Value"]); agg[k][1].add(r["Dispatch_Id"])
        for (kn, cn), (v, ids) in agg.items():
            if "decompress" in kn or "probe" in kn:
                res[f"{os.path.basename(d)}|{kn}|{cn}"] = {"per_launch": v / len(ids), "launches": len(ids)}
json.dump(res, open(os.path.join(o, "pmc_summary.json"), "w"), indent=1)
for k, v in res.items():
    print(k, round(v["per_launch"], 1), v["launches"])
for name in ("ab_open_chunks.jsonl", "ab_ldsall.jsonl", "calib.jsonl"):
    p = os.path.join(o, name)
    if os.path.exists(p):
        for l in open(p):
            print(name, l.strip())
PY
cat "$OUT/rc.txt"
