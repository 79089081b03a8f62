#!/usr/bin/env python3
"""Copy the judged summaries of a scripts/gpu_r2_final.sh session from gpurun_out/<tag>/ into profiles/ (tracked):
bench lines, N sweep, round trips, per-kernel rocprofv3 stats, PMC counters per launch, pmc_traffic*.json (with the
digest of the kernel sources they were measured on: bench.py replays them only for that build)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def counters(path, kernel_substr):
    rows = [r for r in csv.DictReader(open(path)) if kernel_substr in r["Kernel_Name"]]
    launches = len({r["Dispatch_Id"] for r in rows}) or 1
    agg = collections.defaultdict(float)
    for r in rows:
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
    return {k: v / launches for k, v in agg.items()}


def find(src, sub, name):
    hits = glob.glob(os.path.join(src, sub, "**", name), recursive=True)
    return hits[0] if hits else None


def main():
    tag, prefix = sys.argv[1], sys.argv[2]
    import bench

    digest = bench.library_source_digest("lz4")
    src, dst = os.path.join(REPO, "gpurun_out", tag), os.path.join(REPO, "profiles")
    lines = {}
    for name in ("lz4", "snappy", "cascaded", "bitcomp", "ans", "deflate"):
        p = os.path.join(src, f"bench_{name}.json")
        if os.path.exists(p) and os.path.getsize(p):
            lines[name] = json.load(open(p))
    json.dump(lines, open(os.path.join(dst, prefix + "_bench.json"), "w"), indent=1)
    for name in ("nsweep.jsonl", "roundtrip.jsonl", "mortgage_lz4.json"):
        p = os.path.join(src, name)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(dst, f"{prefix}_{name}"))
    for sub, out in (("trace", "lz4"), ("trace_snappy", "snappy"), ("trace_deflate", "deflate")):
        p = find(src, sub, "*kernel_stats.csv")
        if p:
            shutil.copy(p, os.path.join(dst, f"{prefix}_kernel_stats_{out}.csv"))
    pmc = {}
    for name in ("insts", "stall", "fetch", "write"):
        p = find(src, "pmc_" + name, "*counter_collection.csv")
        if p:
            pmc.update(counters(p, "lz4_decompress_window_kernel"))
    if "lz4" in lines:
        cfg = lines["lz4"]["config"]
        pmc["_note"] = (f"per launch of lz4_decompress_window_kernel<checked>, {cfg['chunks_per_gpu']} chunks x 64 KiB; separate "
                        "rocprofv3 --pmc passes; FETCH_SIZE / WRITE_SIZE in KB; lib_source_digest " + digest)
        json.dump(pmc, open(os.path.join(dst, prefix + "_pmc.json"), "w"), indent=1)
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            fetch, write = pmc["FETCH_SIZE"] * 1024, pmc["WRITE_SIZE"] * 1024
            comp = cfg["compressed_bytes_per_gpu"]
            json.dump({
                "algo": "lz4", "dataset": cfg["dataset"], "chunks_per_gpu": cfg["chunks_per_gpu"], "lib_source_digest": digest,
                "hbm_bytes_per_launch": int(fetch + 0.5 * comp + write), "fetch_bytes_counted": int(fetch),
                "write_bytes_counted": int(write), "algorithmic_bytes": lines["lz4"]["roofline"]["algorithmic_bytes_per_launch"],
                "note": "traffic = FETCH_SIZE as counted + the uncounted half of the compressed stream (0.5 x C: gfx950 reports wide "
                        "coalesced reads at 1/2, calibrated in round 1, scripts/gpu_calib.sh) + WRITE_SIZE; separate rocprofv3 --pmc "
                        f"passes, KB units, session {tag}. The far-match gathers (what exceeds half the stream) are tallied at "
                        "64 B per request and left as counted.",
            }, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
    sp = {}
    for name in ("fetch", "write"):
        p = find(src, "pmc_snappy_" + name, "*counter_collection.csv")
        if p:
            sp.update(counters(p, "snappy_decompress_window_kernel"))
    if "snappy" in lines and "FETCH_SIZE" in sp and "WRITE_SIZE" in sp:
        cfg = lines["snappy"]["config"]
        fetch, write = sp["FETCH_SIZE"] * 1024, sp["WRITE_SIZE"] * 1024
        json.dump({
            "algo": "snappy", "dataset": cfg["dataset"], "chunks_per_gpu": cfg["chunks_per_gpu"], "lib_source_digest": digest,
            "hbm_bytes_per_launch": int(fetch + 0.5 * cfg["compressed_bytes_per_gpu"] + write), "fetch_bytes_counted": int(fetch),
            "write_bytes_counted": int(write), "algorithmic_bytes": lines["snappy"]["roofline"]["algorithmic_bytes_per_launch"],
            "note": f"as pmc_traffic.json, for snappy_decompress_window_kernel, session {tag}",
        }, open(os.path.join(dst, "pmc_traffic_snappy.json"), "w"), indent=1)
    dp = {}
    for name in ("insts", "fetch", "write"):
        p = find(src, "pmc_deflate_" + name, "*counter_collection.csv")
        if p:
            dp.update(counters(p, "deflate_decompress_kernel"))
    if "deflate" in lines and dp:
        cfg = lines["deflate"]["config"]
        ddigest = bench.library_source_digest("deflate")
        dp["_note"] = (f"per launch of deflate_decompress_kernel<checked, raw>, {cfg['chunks_per_gpu']} chunks x 64 KiB; separate "
                       "rocprofv3 --pmc passes; FETCH_SIZE / WRITE_SIZE in KB; lib_source_digest " + ddigest)
        json.dump(dp, open(os.path.join(dst, prefix + "_pmc_deflate.json"), "w"), indent=1)
        if "FETCH_SIZE" in dp and "WRITE_SIZE" in dp:
            fetch, write = dp["FETCH_SIZE"] * 1024, dp["WRITE_SIZE"] * 1024
            json.dump({
                "algo": "deflate", "dataset": cfg["dataset"], "chunks_per_gpu": cfg["chunks_per_gpu"], "lib_source_digest": ddigest,
                "hbm_bytes_per_launch": int(fetch + 0.5 * cfg["compressed_bytes_per_gpu"] + write), "fetch_bytes_counted": int(fetch),
                "write_bytes_counted": int(write), "algorithmic_bytes": lines["deflate"]["roofline"]["algorithmic_bytes_per_launch"],
                "note": f"as pmc_traffic.json, for deflate_decompress_kernel, session {tag}",
            }, open(os.path.join(dst, "pmc_traffic_deflate.json"), "w"), indent=1)
    for name in ("pytest_gpu.log", "rc.txt"):
        p = os.path.join(src, name)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(dst, f"{prefix}_{name}"))
    print("collected into profiles/:", sorted(f for f in os.listdir(dst) if f.startswith(prefix) or f.startswith("pmc_traffic")))


if __name__ == "__main__":
    main()
