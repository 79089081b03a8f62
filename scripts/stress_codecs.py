#!/usr/bin/env python3
"""Randomised stress of the Bitcomp and Cascaded codecs on the card, beyond the fixed seeds of the tests (round 6 changed the
Bitcomp compressor's loads and predecessors, the decoder's waits, and the launches of the Cascaded decoder): random element
types, algorithms / schemes, chunk sizes 0 ... 200 KiB at every residue, walks, jumps, constant stretches, zeros and noise
in one chunk; batches of 1 ... 6 000 chunks (Cascaded: every launch shape); compressed bytes compared with the CPU model's,
both decoders' output with the input; random input / output alignments. usage: stress_codecs.py [seeds=20] [first=3000]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
WIDTH = [1, 1, 2, 2, 4, 4, 8, 8]


def column(rng, n_bytes, w):
    dt = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[w]
    n = n_bytes // w + 1
    parts, total = [], 0
    while total < n:
        k = int(rng.choice([1, 7, 64, 300, 2048, 2048 * (4 // w if w < 4 else 1), 5000, 20000]))
        kind = rng.randint(6)
        if kind == 0:
            v = np.cumsum(rng.randint(-3, 4, size=k)).astype(np.int64) + int(rng.randint(0, 1 << 20))
        elif kind == 1:
            v = np.cumsum(rng.randint(-70000, 70000, size=k)).astype(np.int64)
        elif kind == 2:
            v = np.full(k, int(rng.randint(0, 1 << 30)), dtype=np.int64)
        elif kind == 3:
            v = np.zeros(k, dtype=np.int64)
        elif kind == 4:
            v = rng.randint(0, 1 << 62, size=k).astype(np.int64)
        else:
            v = np.repeat(rng.randint(0, 1 << 16, size=k // 3 + 1), rng.randint(1, 6))[:k].astype(np.int64)
        parts.append(v.astype(dt))
        total += k
    return np.concatenate(parts).view(np.uint8)[:n_bytes].copy()


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    import nvcomp_amd
    from oracle import oracle_py as oracle

    oracle.build()
    dev = nvcomp_amd.TorchDevice("cuda:0")
    lib = nvcomp_amd.load_library()
    bad = 0
    n_bc = n_casc = n_casc_chunks = 0
    for seed in range(first, first + seeds):
        rng = np.random.RandomState(seed)
        # ---- Bitcomp ----
        typ, algo = int(rng.randint(8)), int(rng.randint(2))
        w = WIDTH[typ]
        sizes = [int(rng.choice([0, 1, w, 64 * w, 2048 * w, 8192, 65536, int(rng.randint(0, 200000))])) for _ in range(int(rng.randint(1, 40)))]
        chunks = [column(rng, s, w) for s in sizes]
        codec = nvcomp_amd.BatchedCodec(lib, dev, "Bitcomp", (algo, typ))
        comp = codec.compress(chunks, in_align=int(rng.choice([1, 2, 4, 8])))
        n_bc += len(chunks)
        for i, (cc, c) in enumerate(zip(comp, chunks)):
            ref = oracle.bitcomp_compress(c, algo, w)
            if cc.size != ref.size or not np.array_equal(cc, ref):
                bad += 1
                print("seed", seed, "bitcomp", (algo, typ), "chunk", i, "size", c.size, ": compressed bytes differ from the CPU model")
        for checked in (True, False):
            outs, actual, status = codec.decompress(comp, [c.size for c in chunks], checked=checked, comp_align=int(rng.choice([1, 4, 8])),
                                                    out_align=int(rng.choice([1, 2, 8])))
            for i, (o, c) in enumerate(zip(outs, chunks)):
                if not np.array_equal(o, c) or (checked and status[i] != 0):
                    bad += 1
                    print("seed", seed, "bitcomp", (algo, typ), "chunk", i, "size", c.size, "checked", checked, ": decoded bytes differ")
        # ---- Cascaded: a few distinct chunks, cycled to a batch of every launch shape ----
        typ = int(rng.randint(8))
        w = WIDTH[typ]
        sub = int(rng.choice([256, 1024, 4096, 8192, 16384])) // w * w
        opts = (max(sub, 256), typ, int(rng.randint(0, 4)), int(rng.randint(0, 3)), int(rng.randint(2)))
        if opts[0] % w:
            continue
        kinds = [column(rng, int(rng.choice([0, w, 4096, 20000, 65536])) // w * w, w) for _ in range(5)]
        try:
            ccodec = nvcomp_amd.BatchedCodec(lib, dev, "Cascaded", opts)
            ccomp = ccodec.compress(kinds, in_align=8)
        except RuntimeError as e:  # option sets the compressor declines (LDS of the worst case)
            if "returned 11" in str(e):
                continue
            raise
        for i, (cc, c) in enumerate(zip(ccomp, kinds)):
            ref = oracle.cascaded_compress(c, *opts)
            if cc.size != ref.size or not np.array_equal(cc, ref):
                bad += 1
                print("seed", seed, "cascaded", opts, "chunk", i, "size", c.size, ": compressed bytes differ from the CPU model")
        n = int(rng.choice([1, 7, 300, 512, 513, 2000, 4096, 4097, 6000]))
        pick = rng.randint(0, len(kinds), size=n)
        n_casc += 1
        n_casc_chunks += n
        outs, actual, status = ccodec.decompress([ccomp[k] for k in pick], [kinds[k].size for k in pick], comp_align=8, out_align=8)
        for i, k in enumerate(pick):
            if status[i] != 0 or actual[i] != kinds[k].size or not np.array_equal(outs[i], kinds[k]):
                bad += 1
                print("seed", seed, "cascaded", opts, "batch", n, "chunk", i, "kind", k, ": decoded bytes differ / status", status[i])
                break
    print("stress_codecs:", seeds, "seeds:", n_bc, "Bitcomp chunks,", n_casc, "Cascaded batches of", n_casc_chunks, "chunks;", bad, "mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
