#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/gpu_traffic.sh -> one JSON list of per-launch HBM traffic records
(stdout). Every record names the kernel, the workload it ran on and the digest of the kernel sources, which is what
bench.py matches before replaying a number beside a timing."""
import collections
import csv
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

# run name -> [(kernel substring, algo, kind)]
KERNELS = {
    "lz4": [("lz4_decompress_window_kernel", "lz4", "decompress"), ("lz4_compress_wide_kernel", "lz4", "compress")],
    "snappy": [("snappy_decompress_window_kernel", "snappy", "decompress"), ("snappy_compress_wide_kernel", "snappy", "compress")],
    "deflate": [("deflate_decompress_kernel", "deflate", "decompress")],
    "cascaded": [("cascaded_decompress_kernel", "cascaded", "decompress"), ("cascaded_compress_kernel", "cascaded", "compress")],
    "ans": [("ans_decompress_kernel", "ans", "decompress"), ("ans_compress_kernel", "ans", "compress")],
    "bitcomp": [("bitcomp_decompress_kernel", "bitcomp", "decompress"), ("bitcomp_compress_kernel", "bitcomp", "compress")],
    "lz4_mortgage": [("lz4_decompress_window_kernel", "lz4", "decompress")],
    "lz4_mortgage_default": [("lz4_decompress_window_kernel", "lz4", "decompress")],
    "lz4_int32": [("lz4_decompress_window_kernel", "lz4", "decompress")],
    "lz4_mortgage_5120": [("lz4_decompress_window_kernel", "lz4", "decompress")],
    # the batch-size riders of the driver's line
    "lz4_16384": [("lz4_decompress_window_kernel", "lz4", "decompress")],
    "lz4_4096": [("lz4_decompress_pair_kernel", "lz4", "decompress")],
    "lz4_256": [("lz4_decompress_team_kernel", "lz4", "decompress")],
    # the other codecs' own bench lines (python bench.py --algo X at its default size)
    "cascaded_line": [("cascaded_decompress_kernel", "cascaded", "decompress")],
    "bitcomp_line": [("bitcomp_decompress_kernel", "bitcomp", "decompress")],
    "ans_line": [("ans_decompress_kernel", "ans", "decompress")],
    "deflate_line": [("deflate_decompress_kernel", "deflate", "decompress")],
}

# API calls of a run (scripts/gpu_traffic.sh: --steps 2 --warmup 1; the compress leg of bench.py: one call + five timed ones).
# A call may be several dispatches of one kernel (the Cascaded codec runs passes of growing LDS budget): the counters of a
# call are the sum over its dispatches.
CALLS = {"decompress": 3, "compress": 6}


def per_call(path, substr, kind, counter=None):
    rows = [r for r in csv.DictReader(open(path)) if substr in r["Kernel_Name"] and (counter is None or r["Counter_Name"] == counter)]
    if not rows:
        return None
    return sum(float(r["Counter_Value"]) for r in rows) / CALLS[kind]


def main():
    import bench

    out_dir = sys.argv[1]
    records = []
    for run, kernels in KERNELS.items():
        log = os.path.join(out_dir, f"{run}_FETCH_SIZE.log")
        line = None
        if os.path.exists(log):
            for l in open(log):
                if l.startswith("{"):
                    line = json.loads(l)
        if line is None:
            continue
        cfg = line["config"]
        for substr, algo, kind in kernels:
            vals = {}
            for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
                hits = glob.glob(os.path.join(out_dir, f"{run}_{ctr}", "**", "*counter_collection.csv"), recursive=True)
                if hits:
                    vals[ctr] = per_call(hits[0], substr, kind)
            insts = {}
            hits = glob.glob(os.path.join(out_dir, f"{run}_INSTS", "**", "*counter_collection.csv"), recursive=True)
            if hits:
                for ctr in ("SQ_INSTS_VALU", "SQ_INSTS_SALU"):
                    insts[ctr] = per_call(hits[0], substr, kind, ctr)
            if vals.get("FETCH_SIZE") is None or vals.get("WRITE_SIZE") is None:
                continue
            fetch, write = vals["FETCH_SIZE"] * 1024, vals["WRITE_SIZE"] * 1024  # the counters are in KB
            comp, raw = cfg["compressed_bytes_per_gpu"], cfg["uncompressed_bytes_per_gpu"]
            if kind == "compress":  # the compress leg writes its own sizes: the line's extras carry them
                e = line.get("extras", {})
                if not e.get("gpu_compress_ratio"):
                    continue
                comp = int(raw / e["gpu_compress_ratio"])
            # gfx950: FETCH_SIZE = TCC_EA0_RDREQ x 64, and every L2 miss is ONE 128-byte request whatever the access shape --
            # calibrated in round 4 on known load counts (scripts/probes/gather_calib.hip, profiles/r04_feasibility.json:
            # coalesced 16-byte lane loads 0.125 requests per load, lanes 64 bytes apart 0.5, lanes 128 bytes apart and random
            # 16-byte loads 1.0, random byte-aligned 16-byte loads 1.115; TCC_EA0_RDREQ_32B = 0). The factor 2 the guide gives
            # for wide coalesced reads therefore holds for ALL of FETCH_SIZE: the streamed input AND the far-match gathers /
            # candidate probes (a 128-byte line for 4-32 useful bytes). WRITE_SIZE is as counted (it equals the output bytes
            # of every bandwidth-shaped codec).
            uncounted = fetch
            records.append({
                "algo": algo, "kind": kind, "kernel": substr, "dataset": cfg["dataset"], "chunks_per_gpu": cfg["chunks_per_gpu"],
                "producer": cfg.get("producer") if kind == "decompress" else None,
                "lib_source_digest": bench.library_source_digest(algo),
                "fetch_bytes_counted": int(fetch), "write_bytes_counted": int(write),
                "hbm_bytes_per_launch": int(fetch + uncounted + write),
                "valu_wave_insts": insts.get("SQ_INSTS_VALU"), "salu_wave_insts": insts.get("SQ_INSTS_SALU"),
                "algorithmic_bytes": int(comp + raw + (44 if kind == "decompress" else 40) * cfg["chunks_per_gpu"]),
                "calibration": {"fetch_factor": 2.0, "basis": "profiles/r04_feasibility.json fetch_size_calibration: 128-byte "
                                "requests tallied at 64 bytes for every access shape measured", "write_factor": 1.0},
                "note": "2 x FETCH_SIZE + WRITE_SIZE (an ABSOLUTE byte count at the L2's fabric side since the round-4 "
                        "calibration, no longer a lower bound); separate rocprofv3 --pmc passes, KB units; Infinity-Cache "
                        "hits are fabric requests too and are counted",
            })
    print(json.dumps(records, indent=1))


if __name__ == "__main__":
    main()
