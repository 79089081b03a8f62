#!/usr/bin/env python3
"""Randomised stress of the run executor and the run compressor on the card (beyond the fixed seeds of the tests): hand-built
LZ4 blocks and Snappy streams of run sequences -- random periods, literal lengths 0 .. 20, run lengths 4 .. 3 000, merged
and speculated sequences, ordinary sequences in between -- decoded at random output alignments by every launch shape (batch
sizes 1 ... 9 000) and compared with the expansion computed here; columns with random change patterns compressed by the
library and decoded by liblz4 / libsnappy. usage: stress_runs.py [seeds=20] [first=1000]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    import nvcomp_amd
    from nvcomp_amd import _lib
    from oracle import oracle_py as oracle
    from test_lz4_decode import _lz4_block, _lz4_expand

    oracle.build()
    dev = nvcomp_amd.TorchDevice("cuda:0")
    lib = nvcomp_amd.load_library()
    lz4 = nvcomp_amd.BatchedCodec(lib, dev, "LZ4")
    snappy = nvcomp_amd.BatchedCodec(lib, dev, "Snappy")
    bad = 0
    for seed in range(first, first + seeds):
        rng = np.random.RandomState(seed)
        blocks, raws = [], []
        for _ in range(40):
            off = int(rng.choice([1, 2, 4, 8, 16]))
            seqs = [(rng.randint(0, 256, size=off + rng.randint(0, 20)).astype(np.uint8).tobytes(), off, 4 + rng.randint(300))]
            produced = sum(len(l) + m for l, _, m in seqs)
            while produced < 60000:
                kind = rng.randint(12)
                kmax = min(produced // off, 200)
                far = off * int(rng.randint(2, kmax + 1)) if kmax >= 2 else off
                if kind < 6:
                    seqs.append((rng.randint(0, 256, size=int(rng.choice([0, 1, 1, 2, 2, 3, 8, 16, 17, 20]))).astype(np.uint8).tobytes(), off,
                                 int(rng.choice([4, 5, 7, 12, 15, 16, 17, 31, 64, 200, 400, 1000, 3000]))))
                elif kind < 8:
                    seqs.append((rng.randint(0, 256, size=rng.randint(0, 4)).astype(np.uint8).tobytes(), far, 4 + rng.randint(10)))
                    seqs.append((b"", off, 16 + rng.randint(600)))
                elif kind < 9:
                    seqs.append((b"", off, 4 + rng.randint(40)))
                elif kind < 10:
                    off = int(rng.choice([1, 2, 4, 8, 16]))
                    seqs.append((rng.randint(0, 256, size=off).astype(np.uint8).tobytes(), off, 20 + rng.randint(500)))
                elif kind < 11:
                    seqs.append((rng.randint(0, 256, size=rng.randint(0, 30)).astype(np.uint8).tobytes(), 1 + rng.randint(min(produced, 3000)), 4 + rng.randint(60)))
                else:
                    seqs.append((rng.randint(0, 256, size=rng.randint(1, 4)).astype(np.uint8).tobytes(), far, 20 + rng.randint(200)))
                produced = sum(len(l) + m for l, _, m in seqs)
            tail = rng.randint(0, 256, size=5 + rng.randint(10)).astype(np.uint8).tobytes()
            blocks.append(_lz4_block(seqs, tail))
            raws.append(_lz4_expand(seqs, tail))
        for reps in (1, 14, 40, 230):  # 40 ... 9 200 chunks: workgroup per chunk, two waves, persistent waves
            b, r = blocks * reps, raws * reps
            mis = int(rng.randint(16))
            outs, actual, status = lz4.decompress(b, [x.size for x in r], base_misalign=mis)
            ok = (status == 0).all() and all(np.array_equal(o, x) for o, x in zip(outs, r))
            if not ok:
                bad += 1
                print("LZ4 DECODE MISMATCH seed", seed, "reps", reps, "misalign", mis, flush=True)
        # compress: columns with random change patterns
        cols = []
        for _ in range(24):
            width = int(rng.choice([1, 2, 4, 8]))
            n = int(rng.choice([4096, 5000, 20000, 65536, 65535, 65521]))
            nvals = n // width + 2
            runs = rng.randint(1, int(rng.choice([3, 20, 200])), size=nvals)
            vals = np.cumsum(rng.randint(1, 1 << int(rng.choice([4, 12, 30])), size=nvals)).astype(np.uint64)
            col = np.repeat(vals, runs)[: n // width + 1].astype({1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[width]).view(np.uint8)[:n].copy()
            if rng.rand() < 0.3:
                col[rng.randint(0, n, size=rng.randint(1, 400))] ^= 0x55
            cols.append(col)
        for codec, dec in ((lz4, oracle.ref_lz4_decompress), (snappy, oracle.ref_snappy_decompress)):
            comp = codec.compress(cols, in_align=int(rng.choice([1, 16])))
            for i, (cc, c) in enumerate(zip(comp, cols)):
                rc, out = dec(cc, c.size)
                if rc != 0 or not np.array_equal(out, c):
                    bad += 1
                    print("COMPRESS MISMATCH", codec.fmt if hasattr(codec, "fmt") else "?", "seed", seed, "column", i, flush=True)
            outs, actual, status = codec.decompress(comp, [c.size for c in cols])
            if not ((status == 0).all() and all(np.array_equal(o, c) for o, c in zip(outs, cols))):
                bad += 1
                print("ROUND TRIP MISMATCH seed", seed, flush=True)
        print("seed", seed, "done", flush=True)
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
