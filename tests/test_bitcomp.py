"""Bitcomp codec: HIP path (or its host emulation) vs oracle/bitcomp_ref.c.

The reference's Bitcomp bitstream is closed (README.md:13), so parity is pinned to this
library's own stream: compressed bytes must be IDENTICAL to the CPU model's and
decompression must invert both, for both algorithms and every element type the reference's
harness accepts (benchmarks/benchmark_bitcomp_chunked.cu:35-36,47-60,66-101)."""
import ctypes as C

import numpy as np
import pytest

from nvcomp_amd import datasets
from nvcomp_amd._lib import BitcompOpts, NvcompStatus

WIDTH = [1, 1, 2, 2, 4, 4, 8, 8]


def roundtrip(backend, oracle, chunks, algo, typ, comp_align=8, out_align=8):
    codec = backend.codec("Bitcomp", (algo, typ))
    comp = codec.compress(chunks, in_align=8)
    for i, (cc, c) in enumerate(zip(comp, chunks)):
        ref = oracle.bitcomp_compress(c, algo, WIDTH[typ])
        assert cc.size == ref.size and np.array_equal(cc, ref), f"chunk {i}: compressed bytes differ from the CPU model"
        rc, out = oracle.bitcomp_decompress(cc, c.size)
        assert rc == 0 and np.array_equal(out, c)
    outs, actual, status = codec.decompress(comp, [c.size for c in chunks], comp_align=comp_align, out_align=out_align)
    assert (status == NvcompStatus.Success).all(), status
    assert actual.tolist() == [c.size for c in chunks]
    for o, c in zip(outs, chunks):
        assert np.array_equal(o, c)
    sizes = codec.get_decompress_size(comp, comp_align=comp_align)
    assert sizes.tolist() == [c.size for c in chunks]
    # the status-less fast path produces the same bytes
    outs2, actual2, _ = codec.decompress(comp, [c.size for c in chunks], checked=False, comp_align=comp_align,
                                         out_align=out_align)
    assert actual2.tolist() == [c.size for c in chunks]
    for o, c in zip(outs2, chunks):
        assert np.array_equal(o, c)
    return sum(c.size for c in chunks) / max(1, sum(c.size for c in comp))


@pytest.mark.parametrize("typ", range(8))
@pytest.mark.parametrize("algo", [0, 1])
def test_all_types(backend, oracle, algo, typ):
    rng = np.random.RandomState(10 * algo + typ)
    chunks = []
    for name in ("int32", "float32", "lowcard", "zeros", "noise", "text"):
        n = int(rng.randint(1, 2500)) * 8
        chunks.append(datasets.CLASSES[name](n, typ))
    chunks.append(np.zeros(0, dtype=np.uint8))
    chunks.append(datasets.int32_column(65536, 3))
    chunks.append(datasets.noise(8 * 64 * 32 * 2, 1))  # exactly two blocks of 64-bit elements
    roundtrip(backend, oracle, chunks, algo, typ)


def test_ragged_sizes_and_tails(backend, oracle):
    """Sizes around row (64 elements) and block (2048 elements) edges, and sizes that are not a
    multiple of the element size (the harness rejects those, benchmark_bitcomp_chunked.cu:93-100;
    here the odd bytes are carried raw)."""
    rng = np.random.RandomState(7)
    for typ in (1, 3, 5, 7):
        w = WIDTH[typ]
        sizes = [0, 1, w - 1, w, w + 1, 63 * w, 64 * w, 65 * w, 2047 * w, 2048 * w, 2049 * w + 3, 4096 * w + w - 1,
                 5000 * w + 1]
        chunks = []
        for n in sizes:
            walk = np.cumsum(rng.randint(-40, 40, size=n // w + 1)).astype(np.int64)
            chunks.append(walk.astype({1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[w]).view(np.uint8)[:n].copy())
        for algo in (0, 1):
            roundtrip(backend, oracle, chunks, algo, typ)


def test_whole_blocks_take_predecessors_from_the_lane_below(backend, oracle):
    """The compressor requests a whole block's 32 rows at once and takes an element's predecessor from the lane below -- the
    first lane's from the row before, the first row's from the block before (bitcomp.hip.h: encode_chunk). Blocks whose
    first element differs from the last of the block before; a block of one repeated value in between (stored as a marker:
    the block behind it still needs ITS last element); a partial block behind whole ones; every element width, both
    algorithms; walks that wrap around the element's range."""
    rng = np.random.RandomState(42)
    for typ in (0, 1, 2, 3, 4, 5, 6, 7):
        w = WIDTH[typ]
        dt = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[w]
        block = 2048 * (4 // w if w < 4 else 1)
        def walk(n, step):
            return (np.cumsum(rng.randint(-step, step + 1, size=n)).astype(np.int64) + int(rng.randint(0, 1 << 30))).astype(dt)
        chunks = []
        a, b, c = walk(block, 3), walk(block, 40000), walk(block + 77, 9)
        chunks.append(np.concatenate([a, b]).view(np.uint8))                                   # two whole blocks, a jump between them
        chunks.append(np.concatenate([a, np.full(block, a[-1] + 5, dtype=dt), b]).view(np.uint8))  # a constant block in the middle
        chunks.append(np.concatenate([np.full(block, 7, dtype=dt), a, c]).view(np.uint8))      # constant first, partial last
        chunks.append(np.concatenate([np.zeros(block, dtype=dt), np.zeros(block, dtype=dt), b]).view(np.uint8))
        chunks.append(rng.randint(0, 256, size=3 * block * w).astype(np.uint8))                # full-width rows
        for algo in (0, 1):
            roundtrip(backend, oracle, chunks, algo, typ)


def test_unaligned_pointers(backend, oracle):
    chunks = [datasets.int32_column(30000, 2), datasets.float32_column(8192, 1), datasets.lowcard(7777, 3)]
    roundtrip(backend, oracle, chunks, 0, 5, comp_align=1, out_align=1)
    roundtrip(backend, oracle, chunks, 1, 7, comp_align=1, out_align=1)


def test_ratio_on_numeric_columns(backend, oracle):
    """Sorted / slowly varying int32 data: deltas need few bits; sparse data suits algo 1."""
    data = datasets.int32_column(4 * 65536, 5)
    assert roundtrip(backend, oracle, datasets.split_chunks(data), 0, 4) > 2.5
    rng = np.random.RandomState(3)
    sparse = np.zeros(65536, dtype=np.uint32)
    idx = rng.choice(sparse.size, 600, replace=False)
    sparse[idx] = rng.randint(1, 1 << 20, size=idx.size)
    assert roundtrip(backend, oracle, datasets.split_chunks(sparse.view(np.uint8)), 1, 5) > 3
    assert roundtrip(backend, oracle, [np.zeros(65536, np.uint8)], 0, 1) > 100


def test_corrupt_streams(backend, oracle):
    chunks = [datasets.int32_column(30000, 2)] * 7
    codec = backend.codec("Bitcomp", (0, 4))
    comp = codec.compress(chunks)
    rng = np.random.RandomState(5)
    bad = []
    for i, c in enumerate(comp):
        b = c.copy()
        if i == 0:
            b = b[: b.size // 2]
        elif i == 1:
            b[0] ^= 0xFF
        elif i == 2:
            b[12] = 77  # a row width beyond the element width
        elif i == 3:
            b[8] ^= 0x40  # uncompressed size field
        elif i == 4:
            b = b[:7]
        elif i == 5:
            b[5] = 9  # element size code
        bad.append(b)
    caps = [c.size for c in chunks]
    caps[6] -= 4
    outs, actual, status = codec.decompress(bad, caps, comp_align=1, out_align=1)
    for i, (b, cap) in enumerate(zip(bad, caps)):
        rc, ref = oracle.bitcomp_decompress(b, cap)
        if rc == 0:
            assert status[i] == NvcompStatus.Success and np.array_equal(outs[i][: ref.size], ref)
        else:
            assert status[i] == NvcompStatus.ErrorCannotDecompress and actual[i] == 0


def test_opts_validation(backend):
    lib = backend.lib
    out = C.c_size_t(0)
    assert lib.nvcompBatchedBitcompCompressGetMaxOutputChunkSize(65536, BitcompOpts(0, 1), C.byref(out)) == 0
    assert 65536 <= out.value <= 65536 + 65536 // 32 + 4096
    for bad in (BitcompOpts(2, 1), BitcompOpts(-1, 1), BitcompOpts(0, 8), BitcompOpts(0, 0xFF)):
        assert lib.nvcompBatchedBitcompCompressGetMaxOutputChunkSize(65536, bad, C.byref(out)) == NvcompStatus.ErrorInvalidValue
        assert lib.nvcompBatchedBitcompCompressGetTempSize(1, 65536, bad, C.byref(out)) == NvcompStatus.ErrorInvalidValue
    assert lib.nvcompBatchedBitcompCompressGetMaxOutputChunkSize((1 << 24) + 1, BitcompOpts(0, 1), C.byref(out)) \
        == NvcompStatus.ErrorChunkSizeTooLarge


def test_large_chunk(backend, oracle):
    """Chunks well beyond 64 KiB (up to nvcompBitcompCompressionMaxAllowedChunkSize = 16 MiB are accepted)."""
    big = datasets.int32_column((1 << 20) + 4 * 777, 9)
    roundtrip(backend, oracle, [big], 0, 4)
    roundtrip(backend, oracle, [big[: 300000]], 1, 6)
