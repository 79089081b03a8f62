#!/usr/bin/env python3
"""Copy the judged summaries of a scripts/gpu_evidence.sh session from gpurun_out/<tag>/ into profiles/ (tracked): bench
lines, batch-size and data-class sweeps, round trips, per-kernel rocprofv3 stats, PMC counters per launch, the HLIF log and
pmc_traffic_r<NN>.json (keyed by the digest of the kernel sources: bench.py replays it only for that build).
usage: collect_evidence.py <tag> [prefix=r06_final]"""
import json
import os
import shutil
import sys
import glob

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    tag = sys.argv[1]
    prefix = sys.argv[2] if len(sys.argv) > 2 else "r06_final"
    import bench

    src, dst = os.path.join(REPO, "gpurun_out", tag), os.path.join(REPO, "profiles")
    lines = {}
    for name in ("lz4", "snappy", "cascaded", "bitcomp", "ans", "deflate"):
        p = os.path.join(src, f"bench_{name}.json")
        if os.path.exists(p) and os.path.getsize(p):
            lines[name] = json.load(open(p))
    json.dump(lines, open(os.path.join(dst, prefix + "_bench.json"), "w"), indent=1)
    for name in ("nsweep.jsonl", "classes.jsonl", "compress.jsonl", "roundtrip.jsonl", "deflate_1g.json", "rc.txt", "pytest_gpu.log",
                 "calib_mall.jsonl", "kernel_resources.txt"):
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(dst, f"{prefix}_{name}"))
    for name, out in (("compress_phases.jsonl", "compress_phases.jsonl"), ("cascaded_phases.jsonl", "cascaded_phases.jsonl"),
                      ("decode_phases.jsonl", "decode_phases.jsonl"), ("comp/compress_counters.json", "compress_counters.json"),
                      ("comp/roundtrip.jsonl", "gpu_compressed_roundtrip.jsonl"), ("comp/ab_dec.jsonl", "hc_compressed_16384.jsonl"),
                      ("harness/harness.log", "harness.log")):
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(dst, f"{prefix}_{out}"))
    for sub, out in (("trace_cascaded_1g", "cascaded_1gib"), ("trace_cascaded_4g", "cascaded_4gib")):
        hits = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
        if hits:
            shutil.copy(hits[0], os.path.join(dst, f"{prefix}_kernel_stats_{out}.csv"))
    p = os.path.join(src, "hlif", "hlif.log")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{prefix}_hlif.log"))
    for sub, out in (("trace", "lz4"), ("trace_snappy", "snappy"), ("trace_deflate", "deflate"), ("trace_cascaded", "cascaded"),
                     ("trace_bitcomp", "bitcomp"), ("trace_ans", "ans"), ("trace_compress", "lz4_with_compress")):
        hits = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
        if hits:
            shutil.copy(hits[0], os.path.join(dst, f"{prefix}_kernel_stats_{out}.csv"))
    pmc = {"_note": "per launch of the window decoder kernels; separate rocprofv3 --pmc passes (instructions; stalls); "
                    "lz4 / snappy: the headline workload (65 536 chunks x 64 KiB of the mix); lz4_mortgage: 16 384 chunks of the "
                    "mortgage-like key column (liblz4 HC input); lib_source_digest " + bench.library_source_digest("lz4")}
    for sub, key in (("pmc", None), ("pmc_mortgage", "lz4_mortgage")):
        p = os.path.join(src, sub, "pmc.json")
        if os.path.exists(p):
            d = json.load(open(p))
            if key:
                pmc[key] = d.get("lz4") or next(iter(d.values()))
            else:
                pmc.update(d)
    json.dump(pmc, open(os.path.join(dst, prefix + "_pmc.json"), "w"), indent=1)
    p = os.path.join(src, "traffic", bench.PMC_RECORD)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, bench.PMC_RECORD))
    print(sorted(f for f in os.listdir(dst) if f.startswith(prefix) or f == bench.PMC_RECORD))


if __name__ == "__main__":
    main()
