"""Randomised structure fuzz of the LZ4 / Snappy decoders against the oracle:
mixtures of literal runs, short/long matches at near and far distances, periodic
data and chunk sizes that are not multiples of anything."""
import numpy as np
import pytest

from nvcomp_amd import datasets
from nvcomp_amd._lib import NvcompStatus


def synth(rng, size):
    """Concatenate randomly chosen segments: noise, repeats of earlier data at a random
    distance (near or far), short-period runs, and low-entropy bytes."""
    out = np.empty(size, dtype=np.uint8)
    pos = 0
    while pos < size:
        kind = rng.randint(0, 6)
        n = int(min(size - pos, rng.choice([3, 7, 20, 60, 200, 700, 3000, 9000])))
        if kind == 0 or pos < 8:
            out[pos:pos + n] = rng.randint(0, 256, size=n)
        elif kind in (1, 2):  # copy from earlier: near (<= 300 back) or anywhere
            back = rng.randint(1, min(pos, 300) + 1) if kind == 1 else rng.randint(1, pos + 1)
            for i in range(n):  # byte-serial so that back < n makes a period
                out[pos + i] = out[pos + i - back]
        elif kind == 3:
            period = rng.randint(1, 9)
            pat = rng.randint(0, 256, size=period).astype(np.uint8)
            out[pos:pos + n] = np.resize(pat, n)
        elif kind == 4:
            out[pos:pos + n] = rng.randint(0, 4, size=n)
        else:
            out[pos:pos + n] = rng.randint(97, 123, size=n)
        pos += n
    return out


@pytest.mark.parametrize("fmt", ["LZ4", "Snappy"])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzz_structures(backend, lz_path, oracle, fmt, seed):
    rng = np.random.RandomState(seed * 7919 + (0 if fmt == "LZ4" else 1))
    n_chunks = 24 if backend.name == "gpu" else 5
    sizes = [int(rng.choice([100, 1000, 5000, 20000, 65536, 70001, 150000])) for _ in range(n_chunks)]
    chunks = [synth(rng, s) for s in sizes]
    if fmt == "LZ4":
        enc = (lambda c: oracle.ref_lz4_compress(c, int(rng.choice([0, 0, 9])))) if oracle.have_ref() else oracle.lz4_compress
        dec = oracle.lz4_decompress
    else:
        enc = oracle.ref_snappy_compress if oracle.have_ref() else oracle.snappy_compress
        dec = oracle.snappy_decompress
    comp = [enc(c) for c in chunks]
    mis = int(rng.randint(0, 16))
    outs, actual, status = backend.codec(fmt).decompress(comp, sizes, base_misalign=mis)
    assert (status == NvcompStatus.Success).all(), status
    assert actual.tolist() == sizes
    for i, (o, c, cc) in enumerate(zip(outs, chunks, comp)):
        rc, ref = dec(cc, c.size)
        assert rc == 0 and np.array_equal(ref, c)
        assert np.array_equal(o, c), f"chunk {i} (size {c.size}) differs at {int(np.argmax(o != c))}"


@pytest.mark.parametrize("fmt", ["LZ4", "Snappy"])
def test_random_ragged_batches(backend, oracle, fmt):
    """Batches of random size around the launch-shape thresholds of common/lz_launch.hip.h (sixteen-wave teams up to 256
    chunks, eight-wave teams up to 512, two waves per chunk above), chunks of random length, data class, compression level
    and alignment: every byte, size and status as the CPU library has them. (The emulator takes a handful of small
    batches, the GPU thirty of up to 600 chunks.)"""
    if not oracle.have_ref():
        pytest.skip("needs liblz4 / libsnappy (oracle/_ref)")
    rng = np.random.default_rng(11)
    gpu = backend.name == "gpu"
    names = sorted(datasets.CLASSES)
    codec = backend.codec(fmt)
    for it in range(30 if gpu else 3):
        n = int(rng.integers(1, 600)) if gpu else int(rng.integers(1, 6))
        chunks = []
        for _ in range(n):
            size = int(rng.integers(1, 65537 if gpu else 6000))
            d = datasets.CLASSES[names[int(rng.integers(0, len(names)))]](size, int(rng.integers(0, 1000)))
            chunks.append(np.frombuffer(d.tobytes(), dtype=np.uint8)[:size].copy())
        if fmt == "LZ4":
            comp = [oracle.ref_lz4_compress(c, int(rng.integers(0, 13))) for c in chunks]
        else:
            comp = [oracle.ref_snappy_compress(c) for c in chunks]
        outs, act, st = codec.decompress(comp, [c.size for c in chunks], comp_align=int(rng.integers(1, 9)),
                                         out_align=int(rng.integers(1, 9)))
        assert (st == 0).all(), (it, n)
        assert act.tolist() == [c.size for c in chunks]
        for i, (o, c) in enumerate(zip(outs, chunks)):
            assert np.array_equal(o, c), (it, n, i)
