#!/usr/bin/env python3
"""Phase clock of the workgroup-per-chunk LZ4 decoder: a -DNVCOMP_LZW_PROF -DNVCOMP_LZ_TEAM_MAX_BATCH=huge build
(scripts/build_variants.sh teamprof "...") decodes N chunks of the mix; the per-wave cycle sums of every phase are read back.
usage: team_prof.py <lib.so> [chunks ...]"""
import ctypes as C
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
NAMES = ["stage", "flush+build+spec", "wait_A", "enumerate+parse", "wait_B", "prefix+checks", "literals", "register",
         "wait_L", "match_poll", "match_copy", "match_coop", "match_nap", "wait_C", "final_flush", "-"]


def main():
    import torch

    import nvcomp_amd
    from nvcomp_amd import _lib, datasets
    from oracle import oracle_py as oracle

    oracle.build()
    lib = _lib.declare(C.CDLL(os.path.abspath(sys.argv[1])))
    lib.nvcompAmdProfRead.argtypes = [C.c_void_p, C.c_int]
    dev = nvcomp_amd.TorchDevice("cuda:0")
    codec = nvcomp_amd.BatchedCodec(lib, dev, "LZ4")
    data = datasets.silesia_style(64 << 20, 0)
    chunks = datasets.split_chunks(data, 1 << 16)
    _, outs, errs = oracle.batch_run(oracle.LZ4_ENC_HC, chunks, [oracle.lz4_bound(c.size) + 64 for c in chunks],
                                     threads=len(os.sched_getaffinity(0)), use_ref=True)
    comp = [o.copy() for o in outs]
    slots = (C.c_ulonglong * 16)()
    for n in [int(x) for x in sys.argv[2:]] or [256, 1024]:
        sel = [comp[i % len(comp)] for i in range(n)]
        caps = [chunks[i % len(chunks)].size for i in range(n)]
        lib.nvcompAmdProfRead(slots, 16)  # clear
        outs_, sizes, st = codec.decompress(sel, caps, canary=False)
        assert (st == 0).all()
        lib.nvcompAmdProfRead(slots, 16)
        tot = float(sum(slots)) or 1.0
        print(json.dumps({"chunks": n, "cycles_per_chunk_per_wave": round(tot / n / 8), "share": {k: round(v / tot, 4) for k, v in zip(NAMES, slots)}}),
              flush=True)


if __name__ == "__main__":
    main()
