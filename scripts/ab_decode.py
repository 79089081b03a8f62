#!/usr/bin/env python3
"""A/B of library builds on the LZ decode workloads, ONE process: every workload is generated, CPU-compressed and
uploaded once, then every build under --libs (nvcomp_amd/lib/libnvcomp.so + nvcomp_amd/lib/alt/*.so by default) decodes
it: warm-up, timed launches between HIP events, first and last replica byte-compared. One JSON line per (workload, build).

usage: ab_decode.py [--libs a.so b.so ...] [--cases mix,snappy_mix,mortgage,zeros,noise,int32,text] [--mib 4096] [--steps 5]
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
CHUNK = 1 << 16

# case -> (format, dataset, producer, unique MiB, default MiB per launch or None = --mib)
CASES = {
    "mix": ("LZ4", "silesia_style", "hc", 64, None),
    "mix1g": ("LZ4", "silesia_style", "hc", 64, 1024),
    "mix256m": ("LZ4", "silesia_style", "hc", 64, 256),
    "mix128m": ("LZ4", "silesia_style", "hc", 64, 128),
    "mix64m": ("LZ4", "silesia_style", "hc", 64, 64),
    "mix16m": ("LZ4", "silesia_style", "hc", 16, 16),
    "mix512m": ("LZ4", "silesia_style", "hc", 64, 512),
    "snappy256m": ("Snappy", "silesia_style", "snappy", 64, 256),
    "snappy64m": ("Snappy", "silesia_style", "snappy", 64, 64),
    "snappy16m": ("Snappy", "silesia_style", "snappy", 16, 16),
    "snappy32m": ("Snappy", "silesia_style", "snappy", 32, 32),
    "mix32m": ("LZ4", "silesia_style", "hc", 32, 32),
    "mix4m": ("LZ4", "silesia_style", "hc", 4, 4),
    "mix1m": ("LZ4", "silesia_style", "hc", 1, 1),
    "snappy_mix": ("Snappy", "silesia_style", "snappy", 64, None),
    "snappy1g": ("Snappy", "silesia_style", "snappy", 64, 1024),
    "mortgage16m": ("LZ4", "mortgage_col0_like", "fast", 16, 16),
    "mortgage32m": ("LZ4", "mortgage_col0_like", "fast", 32, 32),
    "int32_16m": ("LZ4", "int32", "fast", 16, 16),
    "text16m": ("LZ4", "text", "hc", 16, 16),
    "zeros16m": ("LZ4", "zeros", "fast", 16, 16),
    "snappy_mortgage16m": ("Snappy", "mortgage_col0_like", "snappy", 16, 16),
    "snappy_int32_16m": ("Snappy", "int32", "snappy", 16, 16),
    "snappy_zeros16m": ("Snappy", "zeros", "snappy", 16, 16),
    "snappy_text16m": ("Snappy", "text", "snappy", 16, 16),
    "snappy512m": ("Snappy", "silesia_style", "snappy", 64, 512),
    "mix2g": ("LZ4", "silesia_style", "hc", 64, 2048),
    "mix320m": ("LZ4", "silesia_style", "hc", 64, 320),
    "mortgage": ("LZ4", "mortgage_col0_like", "fast", 64, 1024),
    "mortgage5k": ("LZ4", "mortgage_col0_like", "fast", 64, 314),
    "mortgage2k": ("LZ4", "mortgage_col0_like", "hc", 32, 128),
    "mortgage1k": ("LZ4", "mortgage_col0_like", "hc", 32, 64),
    "mortgage3584": ("LZ4", "mortgage_col0_like", "hc", 32, 224),
    "int32_2k": ("LZ4", "int32", "fast", 32, 128),
    "snappy_int32_2k": ("Snappy", "int32", "snappy", 32, 128),
    "mix224m": ("LZ4", "silesia_style", "hc", 32, 224),
    "mortgage5120": ("LZ4", "mortgage_col0_like", "fast", 32, 320),  # 5 120 chunks: the reference's published run has 5 021 (doc/Benchmarks.md:88-95)
    "mortgage5120_hc": ("LZ4", "mortgage_col0_like", "hc", 32, 320),
    "mortgage_hc": ("LZ4", "mortgage_col0_like", "hc", 64, 1024),
    "snappy_mortgage": ("Snappy", "mortgage_col0_like", "snappy", 64, 1024),
    "zeros": ("LZ4", "zeros", "fast", 16, 1024),
    "noise": ("LZ4", "noise", "fast", 16, 1024),
    "int32": ("LZ4", "int32", "fast", 32, 1024),
    "text": ("LZ4", "text", "hc", 32, 1024),
    "snappy_int32": ("Snappy", "int32", "snappy", 32, 1024),
    "snappy_zeros": ("Snappy", "zeros", "snappy", 16, 1024),
    "snappy_noise": ("Snappy", "noise", "snappy", 16, 1024),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="*", default=None)
    ap.add_argument("--cases", default="mix,snappy_mix")
    ap.add_argument("--mib", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--out", default=None)
    ap.add_argument("--again", action="store_true", help="time the first build once more at the end (run-to-run noise)")
    a = ap.parse_args()
    import torch

    import nvcomp_amd
    from nvcomp_amd import _lib, datasets
    from nvcomp_amd.batched import DeviceBatch
    from oracle import oracle_py as oracle

    oracle.build()
    libs = a.libs or ([_lib.LIB_PATH] + sorted(glob.glob(os.path.join(REPO, "nvcomp_amd", "lib", "alt", "*.so"))))
    handles = [(os.path.basename(p).replace("libnvcomp_", "").replace(".so", ""), _lib.declare(C.CDLL(os.path.abspath(p))))
               for p in libs]
    if a.again:
        handles.append((handles[0][0] + "_again", handles[0][1]))
    dev = nvcomp_amd.TorchDevice("cuda:0")
    threads = len(os.sched_getaffinity(0))
    sink = open(a.out, "a") if a.out else None
    for case in a.cases.split(","):
        if ":" in case:  # ad hoc: format:dataset:producer, 32 MiB unique, 1 GiB per launch
            fmt, ds, producer = case.split(":")
            unique_mib, mib = 32, 1024
        else:
            fmt, ds, producer, unique_mib, mib = CASES[case]
        mib = mib or a.mib
        unique = unique_mib << 20
        gen = getattr(datasets, ds) if hasattr(datasets, ds) else datasets.CLASSES[ds]
        data = gen(unique, 0)
        chunks = datasets.split_chunks(data, CHUNK)
        if fmt == "LZ4":
            codec_id = oracle.LZ4_ENC_HC if producer == "hc" else oracle.LZ4_ENC
            caps = [oracle.lz4_bound(c.size) + 64 for c in chunks]
        else:
            codec_id = oracle.SNAPPY_ENC
            caps = [oracle.snappy_bound(c.size) + 64 for c in chunks]
        _, outs, errs = oracle.batch_run(codec_id, chunks, caps, threads=threads, use_ref=True)
        assert errs == 0
        comp = [o.copy() for o in outs]
        n_u = len(chunks)
        reps = max(1, (mib << 20) // unique)
        n = n_u * reps
        sizes = np.array([c.size for c in comp], dtype=np.uint64)
        offs = np.zeros(n_u, dtype=np.uint64)
        offs[1:] = np.cumsum(sizes)[:-1]
        stride = int(sizes.sum())
        comp_slab = dev.upload(np.concatenate(comp)).repeat(reps)
        out_slab = dev.empty(unique * reps)
        base = dev.upload(data)
        rep = np.arange(reps, dtype=np.uint64)[:, None]
        cptr = (offs[None, :] + rep * np.uint64(stride) + np.uint64(dev.ptr(comp_slab))).reshape(-1)
        optr = (np.arange(n_u, dtype=np.uint64)[None, :] * np.uint64(CHUNK) + rep * np.uint64(unique)
                + np.uint64(dev.ptr(out_slab))).reshape(-1)
        raw_sizes = np.array([c.size for c in chunks], dtype=np.uint64)
        cb = DeviceBatch(comp_slab, dev.upload(cptr.view(np.uint8)), dev.upload(np.tile(sizes, reps).view(np.uint8)), None,
                         np.tile(sizes, reps), n)
        ob = DeviceBatch(out_slab, dev.upload(optr.view(np.uint8)), dev.upload(np.tile(raw_sizes, reps).view(np.uint8)), None,
                         np.tile(raw_sizes, reps), n)
        actual = dev.upload(np.zeros(n, dtype=np.uint64).view(np.uint8))
        statuses = dev.upload(np.full(n, -1, dtype=np.int32).view(np.uint8))
        for tag, lib in handles:
            codec = nvcomp_amd.BatchedCodec(lib, dev, fmt)
            tb = codec.decompress_temp_size(n, CHUNK)
            temp = dev.empty(tb) if tb else None
            out_slab.zero_()
            try:
                for _ in range(a.warmup):
                    assert codec.decompress_async(cb, ob, actual, statuses, temp, tb) == 0
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.steps):
                    codec.decompress_async(cb, ob, actual, statuses, temp, tb)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / a.steps
                st = dev.download(statuses).view(np.int32)[:n]
                ok = bool((st == 0).all()) and all(
                    bool(torch.equal(out_slab[r * unique: (r + 1) * unique], base[:unique])) for r in {0, reps // 2, reps - 1})
                line = {"case": case, "lib": tag, "GBps": round(unique * reps / ms / 1e6, 1), "ms": round(ms, 3), "ok": ok,
                        "chunks": n, "ratio": round(unique / stride, 3)}
            except Exception as e:  # noqa: BLE001
                line = {"case": case, "lib": tag, "error": f"{type(e).__name__}: {e}"[:200]}
            print(json.dumps(line), flush=True)
            if sink:
                sink.write(json.dumps(line) + "\n")
                sink.flush()
        del comp_slab, out_slab, cb, ob
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
