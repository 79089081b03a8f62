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
    "lz4": [("lz4_decompress_window_kernel", "lz4", "decompress"), ("lz4_compress_kernel", "lz4", "compress")],
    "snappy": [("snappy_decompress_window_kernel", "snappy", "decompress")],
    "deflate": [("deflate_decompress_kernel", "deflate", "decompress")],
    "cascaded": [("cascaded_decompress_kernel", "cascaded", "decompress")],
    "lz4_mortgage": [("lz4_decompress_window_kernel", "lz4", "decompress")],
    # the batch-size riders of the driver's line and the unchecked fast path
    "lz4_16384": [("lz4_decompress_window_kernel", "lz4", "decompress")],
    "lz4_4096": [("lz4_decompress_pair_kernel", "lz4", "decompress")],
    "lz4_256": [("lz4_decompress_team_kernel", "lz4", "decompress")],
    "lz4_unchecked": [("lz4_decompress_window_kernel", "lz4", "decompress_unchecked")],
    # the other codecs' own bench lines (python bench.py --algo X at its default size)
    "cascaded_line": [("cascaded_decompress_kernel", "cascaded", "decompress")],
    "bitcomp_line": [("bitcomp_decompress_kernel", "bitcomp", "decompress")],
    "ans_line": [("ans_decompress_kernel", "ans", "decompress")],
    "deflate_line": [("deflate_decompress_kernel", "deflate", "decompress")],
}


# dispatches of the kernel per API call (the Cascaded decoder runs three passes of one kernel: the traffic of a call is their sum)
DISPATCHES_PER_CALL = {"cascaded_decompress_kernel": 3}


def per_launch(path, substr):
    rows = [r for r in csv.DictReader(open(path)) if substr in r["Kernel_Name"]]
    launches = len({r["Dispatch_Id"] for r in rows}) // DISPATCHES_PER_CALL.get(substr, 1)
    total = sum(float(r["Counter_Value"]) for r in rows)
    return (total / launches if launches else None), launches


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
                    vals[ctr], n = per_launch(hits[0], substr)
            if vals.get("FETCH_SIZE") is None or vals.get("WRITE_SIZE") is None:
                continue
            fetch, write = vals["FETCH_SIZE"] * 1024, vals["WRITE_SIZE"] * 1024  # the counters are in KB
            comp, raw = cfg["compressed_bytes_per_gpu"], cfg["uncompressed_bytes_per_gpu"]
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
                "lib_source_digest": bench.library_source_digest(algo),
                "fetch_bytes_counted": int(fetch), "write_bytes_counted": int(write),
                "hbm_bytes_per_launch": int(fetch + uncounted + write),
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
