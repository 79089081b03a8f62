#!/usr/bin/env python3
"""Per-dataset / per-variant decompress sweep in ONE process (saves GPU-box minutes).
Writes one JSON line per case to the given path. Each case = bench.run_case()."""
import argparse
import copy
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "sweep.jsonl"))
    ap.add_argument("--algos", default="lz4,snappy")
    ap.add_argument("--datasets", default="silesia_style,text,table,float_csv,float32,int32,lowcard,zeros,noise")
    ap.add_argument("--mib", type=int, default=512)
    ap.add_argument("--unique-mib", type=int, default=32)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--producer", default="fast")
    ap.add_argument("--unchecked-too", action="store_true")
    ap.add_argument("--dry-run-emu", action="store_true")
    a = ap.parse_args()
    sys.argv = [sys.argv[0]]
    base = bench.parse_args()
    base.dry_run_emu = a.dry_run_emu
    ctx = bench.setup_runtime(base)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "a") as f:
        for algo in a.algos.split(","):
            for ds in a.datasets.split(","):
                for unchecked in ([False, True] if a.unchecked_too else [False]):
                    args = copy.copy(base)
                    args.algo, args.dataset, args.unchecked = algo, ds, unchecked
                    args.mib_per_gpu, args.unique_mib = a.mib, a.unique_mib
                    args.steps, args.warmup, args.producer = a.steps, 1, a.producer
                    args.no_cpu_baseline = True
                    try:
                        r = bench.run_case(args, ctx)
                        line = {"algo": algo, "dataset": ds, "unchecked": unchecked, "GBps": r["value"],
                                "ratio": r["config"]["ratio"], "roofline_frac": r["roofline"]["frac"],
                                "kernel_ms": r["roofline"]["kernel_ms"], "extras": r.get("extras")}
                    except Exception as e:  # keep sweeping; the failure is the data point
                        line = {"algo": algo, "dataset": ds, "unchecked": unchecked, "error": repr(e)}
                    print(json.dumps(line), flush=True)
                    f.write(json.dumps(line) + "\n")
                    f.flush()


if __name__ == "__main__":
    main()
