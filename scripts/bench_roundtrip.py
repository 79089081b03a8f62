#!/usr/bin/env python3
"""GPU compress + decompress round trip of one codec (the shape of the reference's
run_benchmark_template, benchmarks/benchmark_template_chunked.cuh:359-649): prints
ratio, compression and decompression throughput (uncompressed bytes / event time)."""
import argparse
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algo", default="cascaded", choices=["lz4", "snappy", "cascaded", "bitcomp", "ans", "deflate"])
    ap.add_argument("--dataset", default="int32")
    ap.add_argument("--mib", type=int, default=256)
    ap.add_argument("--unique-mib", type=int, default=16)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--chunk", type=int, default=65536)
    ap.add_argument("--opts", default="4096,4,2,1,1", help="cascaded: chunk_size,type,num_RLEs,num_deltas,use_bp; bitcomp: algo,type")
    a = ap.parse_args()
    import torch

    import nvcomp_amd
    from nvcomp_amd import datasets
    from nvcomp_amd.batched import DeviceBatch, empty_batch

    lib = nvcomp_amd.load_library()
    dev = nvcomp_amd.TorchDevice("cuda:0")
    fmt = {"lz4": "LZ4", "snappy": "Snappy", "cascaded": "Cascaded", "bitcomp": "Bitcomp", "ans": "ANS", "deflate": "Deflate"}[a.algo]
    opts = tuple(int(x) for x in a.opts.split(",")) if a.algo in ("cascaded", "bitcomp") else (int(a.opts),) if a.algo == "deflate" and a.opts.isdigit() else None
    codec = nvcomp_amd.BatchedCodec(lib, dev, fmt, opts)
    unique = a.unique_mib << 20
    gen = getattr(datasets, a.dataset) if hasattr(datasets, a.dataset) else datasets.CLASSES[a.dataset]
    data = gen(unique, 0)
    reps = max(1, (a.mib << 20) // unique)
    n_u = unique // a.chunk
    n = n_u * reps
    slab = dev.upload(data).repeat(reps)
    base = dev.ptr(slab)
    ptrs = base + np.arange(n, dtype=np.uint64) * np.uint64(a.chunk)
    sizes = np.full(n, a.chunk, dtype=np.uint64)
    src = DeviceBatch(slab, dev.upload(ptrs.view(np.uint8)), dev.upload(sizes.view(np.uint8)), None, sizes, n)
    max_out = codec.max_compressed_size(a.chunk)
    max_out = (max_out + 7) // 8 * 8
    dst = empty_batch(dev, [max_out] * n, stride=max_out)
    ctb = codec.compress_temp_size(n, a.chunk)
    ctemp = dev.empty(ctb) if ctb else None

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            rc = fn()
            assert rc == 0, rc
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters * 1e-3

    t_c = timed(lambda: codec.compress_async(src, dst, a.chunk, ctemp, ctb))
    if hasattr(lib, "nvcompAmdCompProfRead"):  # phase clocks of a -DNVCOMP_LZM_PROF build (scripts/build_variants.sh)
        import ctypes

        slots = (ctypes.c_ulonglong * 12)()
        if lib.nvcompAmdCompProfRead(slots, 12) > 0:
            tot = float(sum(slots)) or 1.0
            names = ["loop_top", "probe", "insert", "dense_check", "selection", "sizes_scan", "headers", "literals",
                     "cooperative", "tail", "-", "-"]
            print(json.dumps({"phase_share": {k: round(v / tot, 4) for k, v in zip(names, slots)}, "cycles_total": tot}),
                  file=sys.stderr, flush=True)
    csz = dev.download(dst.sizes).view(np.uint64)[:n]
    comp = DeviceBatch(dst.slab, dst.ptrs, dst.sizes, None, csz, n)
    out = empty_batch(dev, [a.chunk] * n, stride=a.chunk)
    actual = dev.upload(np.zeros(n, dtype=np.uint64).view(np.uint8))
    status = dev.upload(np.full(n, -1, dtype=np.int32).view(np.uint8))
    dtb = codec.decompress_temp_size(n, a.chunk)
    dtemp = dev.empty(dtb) if dtb else None
    t_d = timed(lambda: codec.decompress_async(comp, out, actual, status, dtemp, dtb))
    st = dev.download(status).view(np.int32)[:n]
    assert (st == 0).all(), st[:8]
    assert torch.equal(out.slab[: n * a.chunk], slab[: n * a.chunk]), "round trip mismatch"
    total = n * a.chunk
    print(json.dumps({
        "algo": a.algo, "dataset": a.dataset, "opts": opts, "chunks": n, "uncompressed_bytes": total,
        "compressed_bytes": int(csz.sum()), "ratio": round(total / int(csz.sum()), 4),
        "compress_GBps": round(total / t_c / 1e9, 3), "decompress_GBps": round(total / t_d / 1e9, 3),
        "compress_ms": round(t_c * 1e3, 4), "decompress_ms": round(t_d * 1e3, 4), "verified": True}))


if __name__ == "__main__":
    main()
