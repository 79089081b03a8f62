"""nvcompAmdBatchedPackAsync (include/nvcomp/amd_ext.h): chunks spread over worst-case slots -> one contiguous buffer,
offsets by a device-side prefix sum. The building block of bench.py --allgather (benchmarks/benchmark_allgather.cpp
semantics with the ACTUAL compressed bytes on the wire)."""
import numpy as np
import pytest

from nvcomp_amd.batched import make_batch


@pytest.mark.parametrize("stride,mis", [(None, 0), (700, 0), (None, 3), (4104, 5)])
def test_pack_matches_concatenation(backend, stride, mis):
    rng = np.random.RandomState(5)
    sizes = [0, 1, 15, 16, 17, 31, 32, 33, 255, 256, 1023, 1024, 1025, 4099, 0, 7, 600] + [int(rng.randint(0, 650)) for _ in range(300)]
    if stride is not None:
        sizes = [min(s, stride) for s in sizes]
    chunks = [rng.randint(0, 256, size=s).astype(np.uint8) for s in sizes]
    dev, lib = backend.dev, backend.lib
    src = make_batch(dev, chunks, align=8 if stride is None else 1, stride=stride, base_misalign=mis)
    total = sum(sizes)
    packed = dev.upload(np.full(total + 64, 0xEE, dtype=np.uint8))
    offsets = dev.upload(np.zeros(8 * (len(sizes) + 1), dtype=np.uint8))
    rc = lib.nvcompAmdBatchedPackAsync(dev.ptr(src.ptrs), dev.ptr(src.sizes), len(sizes), dev.ptr(packed), total + 64,
                                       dev.ptr(offsets), dev.stream())
    assert rc == 0
    dev.synchronize()
    off = dev.download(offsets).view(np.uint64)
    want = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    assert np.array_equal(off, want)
    got = dev.download(packed)
    assert np.array_equal(got[:total], np.concatenate(chunks)) and (got[total:] == 0xEE).all()


def test_pack_respects_capacity_and_empty_batches(backend):
    dev, lib = backend.dev, backend.lib
    chunks = [np.full(100, 1, np.uint8), np.full(100, 2, np.uint8), np.full(100, 3, np.uint8)]
    src = make_batch(dev, chunks, align=8)
    packed = dev.upload(np.zeros(256, dtype=np.uint8))
    offsets = dev.upload(np.zeros(8 * 4, dtype=np.uint8))
    assert lib.nvcompAmdBatchedPackAsync(dev.ptr(src.ptrs), dev.ptr(src.sizes), 3, dev.ptr(packed), 250, dev.ptr(offsets), dev.stream()) == 0
    dev.synchronize()
    got = dev.download(packed)
    assert dev.download(offsets).view(np.uint64).tolist() == [0, 100, 200, 300]  # the total tells the caller it did not fit
    assert (got[:100] == 1).all() and (got[100:200] == 2).all() and (got[200:] == 0).all()
    assert lib.nvcompAmdBatchedPackAsync(None, None, 0, None, 0, dev.ptr(offsets), dev.stream()) == 0
    dev.synchronize()
    assert dev.download(offsets).view(np.uint64)[0] == 0
    assert lib.nvcompAmdBatchedPackAsync(dev.ptr(src.ptrs), dev.ptr(src.sizes), 3, dev.ptr(packed), 250, None, dev.stream()) == 10
