"""Cascaded scheme pinned to the only vectors the reference holds for it: the three worked examples of
/root/reference doc/cascaded_overview.md (line 9 RLE, line 17 delta, line 25 bit-packing).

The reference's byte stream is closed, so what can be pinned is the SCHEME: the layer streams inside this library's
container must be exactly the runs / values / deltas / (min, packed) of the examples. The container is parsed HERE, in
plain Python (independent of oracle/cascaded_ref.c and of the kernels), for the CPU model and for the HIP compressor."""
import struct

import numpy as np
import pytest

RLE_IN = [3, 9, 9, 4, 4, 4] + [0] * 10 + [1] * 6                       # doc/cascaded_overview.md:9
RLE_VALUES, RLE_RUNS = [3, 9, 4, 0, 1], [1, 2, 3, 10, 6]
DELTA_IN = [15000, 15001, 15002, 15003, 15004, 15204, 15104, 15103, 15102, 15101, 15100]  # :17 and :25
DELTA_OUT = [15000, 1, 1, 1, 1, 200, -100, -1, -1, -1, -1]
BITPACK_MIN, BITPACK_OUT = 15000, [0, 1, 2, 3, 4, 204, 104, 103, 102, 101, 100]


def read_stream(buf, pos, count):
    """u32 bits | u64 min | ceil(count * bits / 32) x u32 -> (bits, min, raw fields, next position)."""
    bits, = struct.unpack_from("<I", buf, pos)
    mn, = struct.unpack_from("<Q", buf, pos + 4)
    words = (count * bits + 31) // 32
    blob = int.from_bytes(buf[pos + 12: pos + 12 + 4 * words], "little")
    fields = [(blob >> (i * bits)) & ((1 << bits) - 1) for i in range(count)] if bits else [0] * count
    return bits, mn, fields, pos + 12 + 4 * words


def parse_single_subchunk(comp):
    """One chunk holding one cascaded (not raw) sub-chunk -> dict of its layer streams."""
    buf = bytes(comp)
    magic, typ, rles, deltas, bp, n_bytes, sub_bytes, num_sub = struct.unpack_from("<IBBBBIII", buf, 0)
    assert magic == 0x43534143 and num_sub == 1
    pos = 20 + 4 * num_sub
    n, = struct.unpack_from("<I", buf, pos)
    assert n != 0xFFFFFFFF, "the example was stored raw: nothing to pin"
    pos += 4
    counts = list(struct.unpack_from("<%dI" % rles, buf, pos))
    pos += 4 * rles
    runs = []
    for l in range(rles):
        bits, mn, f, pos = read_stream(buf, pos, counts[l])
        runs.append({"bits": bits, "min": mn, "fields": f, "values": [x + mn for x in f]})
    c = counts[-1] if rles else n
    bits, mn, f, pos = read_stream(buf, pos, c)
    mask = (1 << 32) - 1
    vals = [(x + mn) & mask for x in f]
    signed = [v - (1 << 32) if v >> 31 else v for v in vals]
    return {"n": n, "counts": counts, "runs": runs, "bits": bits, "min": mn, "fields": f, "values": signed}


def as_bytes(values):
    return np.array(values, dtype=np.int32).view(np.uint8).copy()


def check(name, comp):
    s = parse_single_subchunk(comp(as_bytes(RLE_IN), (4096, 4, 1, 0, 1)))
    assert s["n"] == len(RLE_IN) and s["counts"] == [5], name
    assert s["runs"][0]["values"] == RLE_RUNS and s["values"] == RLE_VALUES, name       # (3,1) (9,2) (4,3) (0,10) (1,6)
    s = parse_single_subchunk(comp(as_bytes(DELTA_IN), (4096, 4, 0, 1, 1)))
    assert s["values"] == DELTA_OUT, name                                                # 15000 1 1 1 1 200 -100 -1 ...
    s = parse_single_subchunk(comp(as_bytes(DELTA_IN), (4096, 4, 0, 0, 1)))
    assert s["min"] == BITPACK_MIN and s["fields"] == BITPACK_OUT and s["bits"] == 8, name  # "requires only 8 bits"


def hip_compressor(be):
    return lambda a, opts: be.codec("Cascaded", opts).compress([a], in_align=8)[0]


def test_overview_examples_cpu(oracle, emu):
    check("oracle", lambda a, opts: oracle.cascaded_compress(a, *opts))
    check("emu", hip_compressor(emu))


@pytest.mark.gpu
def test_overview_examples_gpu(gpu):
    check("gpu", hip_compressor(gpu))
    # and the decoder inverts them
    for values, opts in ((RLE_IN, (4096, 4, 1, 0, 1)), (DELTA_IN, (4096, 4, 0, 1, 1)), (DELTA_IN, (4096, 4, 0, 0, 1))):
        codec = gpu.codec("Cascaded", opts)
        a = as_bytes(values)
        outs, actual, status = codec.decompress(codec.compress([a], in_align=8), [a.size], comp_align=8, out_align=8)
        assert np.array_equal(outs[0], a)
