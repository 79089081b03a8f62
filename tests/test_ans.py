"""ANS codec: HIP path (or its host emulation) vs oracle/ans_ref.c.

The reference's ANS bitstream is closed (README.md:10,17), so parity is pinned to this
library's own stream: compressed bytes must be IDENTICAL to the CPU model's and
decompression must invert both (benchmarks/benchmark_ans_chunked.cu:29-81: byte data, a
single format type, round trip verified by benchmark_template_chunked.cuh:553-584)."""
import ctypes as C

import numpy as np
import pytest

from nvcomp_amd import datasets
from nvcomp_amd._lib import ANSOpts, NvcompStatus


def roundtrip(backend, oracle, chunks, comp_align=8, out_align=8):
    codec = backend.codec("ANS")
    comp = codec.compress(chunks, in_align=8)
    for i, (cc, c) in enumerate(zip(comp, chunks)):
        ref = oracle.ans_compress(c)
        assert cc.size == ref.size and np.array_equal(cc, ref), f"chunk {i}: compressed bytes differ from the CPU model"
        assert cc.size <= c.size + 12
        rc, out = oracle.ans_decompress(cc, c.size)
        assert rc == 0 and np.array_equal(out, c)
    for checked in (True, False):
        outs, actual, status = codec.decompress(comp, [c.size for c in chunks], checked=checked, comp_align=comp_align,
                                                out_align=out_align)
        if checked:
            assert (status == NvcompStatus.Success).all(), status
        assert actual.tolist() == [c.size for c in chunks]
        for o, c in zip(outs, chunks):
            assert np.array_equal(o, c)
    sizes = codec.get_decompress_size(comp, comp_align=comp_align)
    assert sizes.tolist() == [c.size for c in chunks]
    return sum(c.size for c in chunks) / max(1, sum(c.size for c in comp))


@pytest.mark.parametrize("name", sorted(datasets.CLASSES))
def test_dataset_classes(backend, oracle, name):
    data = datasets.CLASSES[name](3 * 65536 + 12345, 4)
    ratio = roundtrip(backend, oracle, datasets.split_chunks(data))
    if name in ("text", "table", "float_csv", "lowcard", "zeros"):
        assert ratio > 1.3, ratio
    if name == "noise":
        assert 0.99 < ratio <= 1.0


def test_ragged_sizes(backend, oracle):
    rng = np.random.RandomState(11)
    sizes = [0, 1, 3, 255, 256, 257, 1023, 1024, 1025, 1279, 1280, 1281, 4095, 4096 + 17, 10000, 65535, 65536, 70001]
    chunks = []
    for n in sizes:
        # skewed bytes so that coding wins whenever it is allowed to
        chunks.append(np.minimum(rng.geometric(0.3, size=n), 255).astype(np.uint8))
    roundtrip(backend, oracle, chunks)
    roundtrip(backend, oracle, chunks, comp_align=1, out_align=1)


def test_extreme_histograms(backend, oracle):
    rng = np.random.RandomState(2)
    one = np.full(65536, 7, dtype=np.uint8)                       # a single symbol: freq 1024, no words at all
    two = (rng.rand(65536) < 0.001).astype(np.uint8) * 200        # one dominant symbol and a rare one
    rare = np.zeros(65536, dtype=np.uint8)
    rare[rng.choice(65536, 255, replace=False)] = np.arange(1, 256, dtype=np.uint8)  # 255 symbols seen once each
    flat = np.tile(np.arange(256, dtype=np.uint8), 256)           # exactly uniform
    ramp = np.repeat(np.arange(256, dtype=np.uint8), np.arange(256) + 1)[:65536].copy()
    assert roundtrip(backend, oracle, [one, two, rare]) > 20
    roundtrip(backend, oracle, [flat, ramp])


def test_corrupt_streams(backend, oracle):
    chunks = [datasets.text(65536, 3)] * 8
    codec = backend.codec("ANS")
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
            b[rng.randint(1040, b.size)] ^= 0x10  # a stream word
        elif i == 3:
            b[4] ^= 0x40  # uncompressed size field
        elif i == 4:
            b[16] ^= 0x01  # a frequency: the table no longer sums to 1024
        elif i == 5:
            b[530] ^= 0x20  # a final state
        elif i == 6:
            b[12] += 1  # word count
        bad.append(b)
    caps = [c.size for c in chunks]
    outs, actual, status = codec.decompress(bad, caps, comp_align=1, out_align=1)
    for i, (b, cap) in enumerate(zip(bad, caps)):
        rc, ref = oracle.ans_decompress(b, cap)
        if rc == 0:
            assert status[i] == NvcompStatus.Success and np.array_equal(outs[i][: ref.size], ref)
        else:
            assert status[i] == NvcompStatus.ErrorCannotDecompress and actual[i] == 0
    assert (status[:7] != 0).all()  # every one of these corruptions is detectable
    small = codec.decompress(comp[:1], [100])
    assert small[2][0] == NvcompStatus.ErrorCannotDecompress and small[1][0] == 0


def test_opts_validation(backend):
    lib = backend.lib
    out = C.c_size_t(0)
    assert lib.nvcompBatchedANSCompressGetMaxOutputChunkSize(65536, ANSOpts(0), C.byref(out)) == 0
    assert 65536 <= out.value <= 65536 + 64
    assert lib.nvcompBatchedANSCompressGetMaxOutputChunkSize(65536, ANSOpts(1), C.byref(out)) == NvcompStatus.ErrorInvalidValue
    assert lib.nvcompBatchedANSCompressGetTempSize(1, 65536, ANSOpts(3), C.byref(out)) == NvcompStatus.ErrorInvalidValue
    assert lib.nvcompBatchedANSCompressGetMaxOutputChunkSize((1 << 24) + 1, ANSOpts(0), C.byref(out)) \
        == NvcompStatus.ErrorChunkSizeTooLarge


def test_large_chunk(backend, oracle):
    """Chunks well beyond 64 KiB (the API allows up to nvcompANSCompressionMaxAllowedChunkSize = 16 MiB)."""
    big = datasets.text((1 << 20) + 12345, 9)
    roundtrip(backend, oracle, [big, datasets.lowcard(300000, 1)])
