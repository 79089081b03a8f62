"""The order in which the persistent waves of the batched LZ decoders take a batch's chunks (common/lz_order.hip.h,
nvcompAmdBatched<Fmt>DecompressOrderAsync): a permutation, the expensive chunks -- many short sequences -- first."""
import numpy as np
import pytest

from nvcomp_amd import datasets
from nvcomp_amd.batched import make_batch


@pytest.mark.parametrize("fmt", ["LZ4", "Snappy"])
def test_expensive_chunks_first(backend, oracle, fmt):
    names = ["zeros", "text", "noise", "int32", "table", "lowcard", "text", "zeros", "float_csv", "noise"]
    chunks, kinds = [], []
    for rep in range(3):
        for j, nme in enumerate(names):
            chunks.append(datasets.CLASSES[nme](65536, rep * 16 + j))
            kinds.append(nme)
    chunks.append(np.zeros(0, np.uint8))  # an empty chunk is a (cheap) chunk too
    kinds.append("empty")
    if fmt == "LZ4":
        comp = [oracle.ref_lz4_compress(c) if oracle.have_ref() else oracle.lz4_compress(c) for c in chunks]
    else:
        comp = [oracle.ref_snappy_compress(c) if oracle.have_ref() else oracle.snappy_compress(c) for c in chunks]
    dev, lib = backend.dev, backend.lib
    n = len(comp)
    batch = make_batch(dev, comp, align=1)
    tb = 256 + ((5 * n + 15) & ~15)
    temp = dev.empty(tb)
    order = dev.upload(np.full(n, 0xFFFFFFFF, dtype=np.uint32).view(np.uint8))
    classes = dev.upload(np.full(n, 0xFF, dtype=np.uint8))
    fn = getattr(lib, f"nvcompAmdBatched{fmt}DecompressOrderAsync")
    assert fn(dev.ptr(batch.ptrs), dev.ptr(batch.sizes), n, dev.ptr(temp), tb, dev.ptr(order), dev.ptr(classes), dev.stream()) == 0
    dev.synchronize()
    o = dev.download(order).view(np.uint32)[:n]
    c = dev.download(classes)[:n]
    assert sorted(o.tolist()) == list(range(n)), "every chunk exactly once"
    assert (np.diff(c[o].astype(int)) <= 0).all(), "classes in descending order"
    place = {k: [int(np.nonzero(o == i)[0][0]) for i, kk in enumerate(kinds) if kk == k] for k in set(kinds)}
    # text and table rows (thousands of short sequences) come before incompressible chunks (one literal run), which
    # come before runs (one match)
    assert max(place["text"]) < min(place["noise"]) and max(place["table"]) < min(place["noise"])
    if fmt == "LZ4":  # (a run is ONE LZ4 sequence, but a train of a thousand 64-byte Snappy copies: no cheaper than noise there)
        assert max(place["noise"]) < min(place["zeros"])
    assert place["empty"][0] >= max(place["text"])
    # a temp buffer that is too small is refused, not overrun
    assert fn(dev.ptr(batch.ptrs), dev.ptr(batch.sizes), n, dev.ptr(temp), tb - 16, dev.ptr(order), dev.ptr(classes), dev.stream()) != 0
