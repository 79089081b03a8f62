#!/usr/bin/env python3
"""A/B of library builds on the LZ COMPRESS workloads, one process (see ab_decode.py): GB/s, ratio, and a round trip
through the same build's decoder compared with the input."""
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
CASES = {"mix": ("LZ4", "silesia_style", 64, 4096), "snappy_mix": ("Snappy", "silesia_style", 64, 4096),
         "mix1g": ("LZ4", "silesia_style", 64, 1024), "int32": ("LZ4", "int32", 32, 1024), "text": ("LZ4", "text", 32, 1024),
         "mortgage": ("LZ4", "mortgage_col0_like", 64, 1024), "noise": ("LZ4", "noise", 16, 1024),
         "zeros": ("LZ4", "zeros", 16, 1024), "snappy_int32": ("Snappy", "int32", 32, 1024),
         "snappy_mortgage": ("Snappy", "mortgage_col0_like", 64, 1024)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="*", default=None)
    ap.add_argument("--cases", default="mix,snappy_mix")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--out", default=None)
    ap.add_argument("--prof", action="store_true", help="read the -DNVCOMP_LZM_PROF phase clocks of builds that have them")
    a = ap.parse_args()
    import torch

    import nvcomp_amd
    from nvcomp_amd import _lib, datasets
    from nvcomp_amd.batched import DeviceBatch, empty_batch

    libs = a.libs or ([_lib.LIB_PATH] + sorted(glob.glob(os.path.join(REPO, "nvcomp_amd", "lib", "alt", "*.so"))))
    handles = [(os.path.basename(p).replace("libnvcomp_", "").replace(".so", ""), _lib.declare(C.CDLL(os.path.abspath(p))))
               for p in libs]
    dev = nvcomp_amd.TorchDevice("cuda:0")
    sink = open(a.out, "a") if a.out else None
    for case in a.cases.split(","):
        fmt, ds, unique_mib, mib = CASES[case]
        unique = unique_mib << 20
        gen = getattr(datasets, ds) if hasattr(datasets, ds) else datasets.CLASSES[ds]
        data = gen(unique, 0)
        reps = max(1, (mib << 20) // unique)
        n = unique // CHUNK * reps
        slab = dev.upload(data).repeat(reps)
        ptrs = dev.ptr(slab) + np.arange(n, dtype=np.uint64) * np.uint64(CHUNK)
        sizes = np.full(n, CHUNK, dtype=np.uint64)
        src = DeviceBatch(slab, dev.upload(ptrs.view(np.uint8)), dev.upload(sizes.view(np.uint8)), None, sizes, n)
        for tag, lib in handles:
            codec = nvcomp_amd.BatchedCodec(lib, dev, fmt)
            max_out = (codec.max_compressed_size(CHUNK) + 7) // 8 * 8
            dst = empty_batch(dev, [max_out] * n, stride=max_out)
            tb = codec.compress_temp_size(n, CHUNK)
            temp = dev.empty(tb) if tb else None
            try:
                assert codec.compress_async(src, dst, CHUNK, temp, tb) == 0
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.steps):
                    codec.compress_async(src, dst, CHUNK, temp, tb)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / a.steps
                csz = dev.download(dst.sizes).view(np.uint64)[:n]
                # round trip through the same build's decoder
                out = dev.empty(unique * reps)
                optr = dev.ptr(out) + np.arange(n, dtype=np.uint64) * np.uint64(CHUNK)
                ob = DeviceBatch(out, dev.upload(optr.view(np.uint8)), dev.upload(sizes.view(np.uint8)), None, sizes, n)
                cb = DeviceBatch(dst.slab, dst.ptrs, dst.sizes, None, csz, n)
                st = dev.upload(np.full(n, -1, dtype=np.int32).view(np.uint8))
                dtb = codec.decompress_temp_size(n, CHUNK)
                dtemp = dev.empty(dtb) if dtb else None
                assert codec.decompress_async(cb, ob, None, st, dtemp, dtb) == 0
                torch.cuda.synchronize()
                ok = bool((dev.download(st).view(np.int32)[:n] == 0).all()) and bool(torch.equal(out, slab))
                line = {"case": case, "lib": tag, "GBps": round(unique * reps / ms / 1e6, 1), "ms": round(ms, 3),
                        "ratio": round(unique * reps / float(csz.sum()), 4), "ok": ok}
                if a.prof and hasattr(lib, "nvcompAmdCompProfRead"):
                    slots = (C.c_ulonglong * 12)()
                    if lib.nvcompAmdCompProfRead(slots, 12) > 0:
                        tot = float(sum(slots)) or 1.0
                        names = ["loop_top", "probe", "insert", "dense", "select", "scan", "headers", "literals", "coop", "tail", "-", "-"]
                        if tag.startswith("w") or tag == "libnvcomp":  # the 256-position steps (common/lz_match_wide.hip.h)
                            names = ["loop_top", "probe_next_step", "hits", "dense", "queue_and_requests", "measure", "select",
                                     "emit", "coop", "tail", "-", "-"]
                        line["phase_share"] = {k: round(v / tot, 3) for k, v in zip(names, slots) if v}
                del out, ob
            except Exception as e:  # noqa: BLE001
                line = {"case": case, "lib": tag, "error": f"{type(e).__name__}: {e}"[:200]}
            print(json.dumps(line), flush=True)
            if sink:
                sink.write(json.dumps(line) + "\n")
                sink.flush()
            del dst
            torch.cuda.empty_cache()
        del slab, src
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
