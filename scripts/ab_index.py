#!/usr/bin/env python3
"""The token index alone (nvcompAmdBatched{LZ4,Snappy}TokenIndexAsync: one wave per chunk, nothing else on the card): time per
launch over the headline batch, for every library build given. usage: ab_index.py --libs a.so b.so [--mib 4096] [--algo lz4]"""
import argparse, ctypes as C, json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
ap = argparse.ArgumentParser()
ap.add_argument("--libs", nargs="+", required=True)
ap.add_argument("--mib", type=int, default=2048)
ap.add_argument("--algo", default="lz4")
ap.add_argument("--dataset", default="silesia_style")
ap.add_argument("--steps", type=int, default=5)
args = ap.parse_args()
import torch
import nvcomp_amd
from nvcomp_amd import datasets, _lib
from oracle import oracle_py as oracle
oracle.build()
dev = nvcomp_amd.TorchDevice("cuda:0")
unique = 64 << 20
data = getattr(datasets, args.dataset)(unique, 0)
chunks = datasets.split_chunks(data)
from concurrent.futures import ThreadPoolExecutor
enc = (lambda c: oracle.ref_lz4_compress(c, 12)) if args.algo == "lz4" else oracle.ref_snappy_compress
with ThreadPoolExecutor(64) as ex:
    comp = list(ex.map(enc, chunks))
reps = (args.mib << 20) // unique
sizes = np.array([c.size for c in comp], dtype=np.uint64)
offs = np.zeros(len(comp), dtype=np.uint64); offs[1:] = np.cumsum(sizes)[:-1]
host = np.concatenate(comp)
slab = dev.upload(np.tile(host, reps))
n = len(comp) * reps
stride = int(sizes.sum())
ptrs = (offs[None, :] + (np.arange(reps, dtype=np.uint64) * np.uint64(stride))[:, None] + np.uint64(dev.ptr(slab))).reshape(-1)
d_ptrs = dev.upload(ptrs.view(np.uint8)); d_sizes = dev.upload(np.tile(sizes, reps).view(np.uint8))
lists = dev.empty(n * 64 * 344 * 2); info = dev.empty(n * 8)
fmt = "LZ4" if args.algo == "lz4" else "Snappy"
for path in args.libs:
    lib = _lib.declare(C.CDLL(path))
    fn = getattr(lib, f"nvcompAmdBatched{fmt}TokenIndexAsync")
    for _ in range(2):
        assert fn(dev.ptr(d_ptrs), dev.ptr(d_sizes), n, dev.ptr(lists), dev.ptr(info), dev.stream()) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        fn(dev.ptr(d_ptrs), dev.ptr(d_sizes), n, dev.ptr(lists), dev.ptr(info), dev.stream())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    inf = dev.download(info).view(np.uint32).reshape(n, 2)
    print(json.dumps({"lib": os.path.basename(path), "algo": args.algo, "chunks": n, "ms": round(ms, 3),
                      "tokens_indexed": int(inf[:, 0].astype(np.int64).sum()), "us_per_chunk_wave": None}), flush=True)
