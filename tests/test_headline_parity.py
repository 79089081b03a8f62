"""BASELINE.json configs[1] at the size the numbers are quoted on: 16 384 chunks of 64 KiB of the Silesia-style mix,
compressed on the host by liblz4's LZ4_compress_HC(12) (what examples/lz4_cpu_compression.cu:59-74 calls) / libsnappy,
decoded by nvcompBatched{LZ4,Snappy}DecompressAsync with statuses and sizes requested -- and every output byte compared
with what LZ4_decompress_safe / snappy::RawUncompress wrote for the same stream (the interoperability pin of
examples/lz4_cpu_compression.cu:121-137), not only with the originals. The other GPU parity tests run a few chunks per
class through every launch shape; this one runs the persistent-wave launch at the batch size of the bench line."""
import os

import numpy as np
import pytest

from nvcomp_amd import datasets

CHUNK = 1 << 16
UNIQUE_MIB = 64
REPLICAS = 16  # 16 x 1 024 chunks = 16 384 chunks, 1 GiB of output, every replica in device memory of its own


# (format, dataset, producer): the headline mix; and, since round 6, the columns the run executor takes (common/lz_window.hip.h:
# execute_run_batch) -- the sorted-key column through liblz4 HC (clean runs) and through its default compressor (every run
# starts with a match from far back: the speculated matches), the int32 column -- at the same batch size
CASES = [("LZ4", "silesia_style", "hc"), ("Snappy", "silesia_style", "snappy"), ("LZ4", "mortgage_col0_like", "hc"),
         ("LZ4", "mortgage_col0_like", "fast"), ("LZ4", "int32", "fast"), ("Snappy", "int32", "snappy")]


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,dataset,producer", CASES, ids=[f"{f}-{d}-{p}" for f, d, p in CASES])
def test_headline_batch_against_the_cpu_decoder(gpu, oracle, fmt, dataset, producer):
    import torch

    from nvcomp_amd.batched import BatchedCodec, DeviceBatch

    if not oracle.have_ref():
        pytest.skip("oracle/_ref (liblz4 / libsnappy shim) is not built")
    dev = gpu.dev
    threads = len(os.sched_getaffinity(0))
    gen = getattr(datasets, dataset) if hasattr(datasets, dataset) else datasets.CLASSES[dataset]
    data = gen(UNIQUE_MIB << 20, 0)
    chunks = datasets.split_chunks(data, CHUNK)
    n_u = len(chunks)
    if fmt == "LZ4":
        enc, dec = (oracle.LZ4_ENC_HC if producer == "hc" else oracle.LZ4_ENC), oracle.LZ4_DEC
        caps = [oracle.lz4_bound(c.size) + 64 for c in chunks]
    else:
        enc, dec = oracle.SNAPPY_ENC, oracle.SNAPPY_DEC
        caps = [oracle.snappy_bound(c.size) + 64 for c in chunks]
    _, comp, errs = oracle.batch_run(enc, chunks, caps, threads=threads, use_ref=True)
    assert errs == 0
    comp = [c.copy() for c in comp]
    # the checker: the CPU library's decoder on the same streams
    _, ref_out, errs = oracle.batch_run(dec, comp, [c.size for c in chunks], threads=threads, use_ref=True)
    assert errs == 0 and all(o.size == c.size for o, c in zip(ref_out, chunks))
    ref = np.concatenate(ref_out)
    assert np.array_equal(ref, data), "the CPU decoder does not restore the originals: the producer is broken"

    n = n_u * REPLICAS
    sizes = np.array([c.size for c in comp], dtype=np.uint64)
    offs = np.zeros(n_u, dtype=np.uint64)
    offs[1:] = np.cumsum(sizes)[:-1]
    stride = int(sizes.sum())
    comp_slab = dev.upload(np.concatenate(comp)).repeat(REPLICAS)
    out_slab = dev.empty(data.size * REPLICAS)
    rep = (np.arange(REPLICAS, dtype=np.uint64))[:, None]
    comp_ptrs = (offs[None, :] + rep * np.uint64(stride) + np.uint64(dev.ptr(comp_slab))).reshape(-1)
    out_ptrs = (np.arange(n_u, dtype=np.uint64)[None, :] * np.uint64(CHUNK) + rep * np.uint64(data.size)
                + np.uint64(dev.ptr(out_slab))).reshape(-1)
    raw_sizes = np.tile(np.array([c.size for c in chunks], dtype=np.uint64), REPLICAS)
    cb = DeviceBatch(comp_slab, dev.upload(comp_ptrs.view(np.uint8)), dev.upload(np.tile(sizes, REPLICAS).view(np.uint8)),
                     None, np.tile(sizes, REPLICAS), n)
    ob = DeviceBatch(out_slab, dev.upload(out_ptrs.view(np.uint8)), dev.upload(raw_sizes.view(np.uint8)), None, raw_sizes, n)
    actual = dev.upload(np.zeros(n, dtype=np.uint64).view(np.uint8))
    statuses = dev.upload(np.full(n, -1, dtype=np.int32).view(np.uint8))
    codec = BatchedCodec(gpu.lib, dev, fmt)
    tb = codec.decompress_temp_size(n, CHUNK)
    temp = dev.empty(tb) if tb else None
    for _ in range(2):  # the second call reuses the temp buffer (the ticket counter of the persistent launch)
        out_slab.zero_()
        assert codec.decompress_async(cb, ob, actual, statuses, temp, tb) == 0
        dev.synchronize()
        st = dev.download(statuses).view(np.int32)[:n]
        assert (st == 0).all(), f"{int((st != 0).sum())} of {n} chunks failed"
        assert np.array_equal(dev.download(actual).view(np.uint64)[:n], raw_sizes)
        ref_dev = dev.upload(ref)
        for r in range(REPLICAS):
            assert torch.equal(out_slab[r * data.size: (r + 1) * data.size], ref_dev[: data.size]), f"replica {r} differs from the CPU decoder's output"
        del ref_dev
