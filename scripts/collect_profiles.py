#!/usr/bin/env python3
"""Copy the judged summaries of a scripts/gpu_final.sh session from gpurun_out/<tag>/ into profiles/
(tracked): bench lines, sweeps, per-kernel rocprofv3 stats, PMC counters per launch, pmc_traffic.json."""
import collections
import csv
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pmc_per_launch(path, kernel_substr, launches):
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if kernel_substr in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
    return {k: v / launches for k, v in agg.items()}


def main():
    tag, prefix = sys.argv[1], sys.argv[2]
    src = os.path.join(REPO, "gpurun_out", tag)
    dst = os.path.join(REPO, "profiles")
    os.makedirs(dst, exist_ok=True)
    bench = {}
    for name in ("bench_lz4", "bench_lz4_unchecked", "bench_lz4_direct", "bench_lz4_serial", "bench_snappy"):
        p = os.path.join(src, name + ".json")
        if os.path.exists(p):
            bench[name] = json.load(open(p))
    json.dump(bench, open(os.path.join(dst, prefix + "_bench.json"), "w"), indent=1)
    for name in ("sweep.jsonl", "roundtrip.jsonl"):
        p = os.path.join(src, name)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(dst, f"{prefix}_{name}"))
    for d, out in (("trace", "lz4"), ("trace_snappy", "snappy"), ("trace_ans", "ans"), ("trace_bitcomp", "bitcomp"),
                   ("trace_cascaded", "cascaded")):
        p = os.path.join(src, d, "r_kernel_stats.csv")
        if os.path.exists(p):
            shutil.copy(p, os.path.join(dst, f"{prefix}_kernel_stats_{out}.csv"))
    # PMC passes ran `bench.py --steps 2 --warmup 1`: 1 warm-up + 2 timed launches + 1 verification launch
    pmc = {}
    for name in ("insts", "stall", "fetch", "write"):
        p = os.path.join(src, "pmc_" + name, "r_counter_collection.csv")
        if os.path.exists(p):
            rows = list(csv.DictReader(open(p)))
            launches = len({r["Dispatch_Id"] for r in rows if "lz4_decompress_window_kernel" in r["Kernel_Name"]})
            pmc.update(pmc_per_launch(p, "lz4_decompress_window_kernel", max(1, launches)))
    pmc["_note"] = ("per launch of lz4_decompress_window_kernel<checked>, 16384 chunks x 64 KiB (1 GiB out, 471 MB in); "
                    "separate rocprofv3 --pmc passes; FETCH_SIZE/WRITE_SIZE in KB")
    json.dump(pmc, open(os.path.join(dst, prefix + "_pmc.json"), "w"), indent=1)
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        fetch, write = pmc["FETCH_SIZE"] * 1024, pmc["WRITE_SIZE"] * 1024
        json.dump({
            "algo": "lz4", "dataset": "silesia_style", "chunks_per_gpu": 16384,
            "hbm_bytes_per_launch": int(fetch + write), "fetch_bytes": int(fetch), "write_bytes": int(write),
            "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB units) of lz4_decompress_window_kernel, "
                    f"session {tag}; FETCH_SIZE is NOT doubled: the 2x gfx950 correction of MI355X_MICROARCH.md applies to "
                    "wide coalesced streaming reads, this kernel reads mostly scattered 4-byte gathers and 16-byte lane loads",
        }, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
    print("collected", sorted(os.listdir(dst)))


if __name__ == "__main__":
    main()
