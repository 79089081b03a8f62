"""Snappy batched codec parity: HIP path (or its host emulation) vs the CPU oracle
and libsnappy. The reference only round-trips Snappy
(benchmarks/benchmark_snappy_synth.cpp:286-295); BASELINE.json's north_star adds
bit-exactness against the snappy CPU decoder, and CHANGELOG.md:182-184 requires
that legal streams its own compressor never emits still decode."""
import numpy as np
import pytest

from nvcomp_amd import datasets
from nvcomp_amd._lib import NvcompStatus


def cpu_compress(oracle, chunks):
    if oracle.have_ref():
        return [oracle.ref_snappy_compress(c) for c in chunks]
    return [oracle.snappy_compress(c) for c in chunks]


def check_decode(backend, oracle, chunks, comp, **kw):
    codec = backend.codec("Snappy")
    caps = [c.size for c in chunks]
    outs, actual, status = codec.decompress(comp, caps, **kw)
    if status is not None:
        assert (status == NvcompStatus.Success).all(), status
    if actual is not None:
        assert actual.tolist() == caps
    for i, (o, c, cc) in enumerate(zip(outs, chunks, comp)):
        assert np.array_equal(o, c), f"chunk {i} differs"
        rc, ref = oracle.snappy_decompress(cc, c.size)
        assert rc == 0 and np.array_equal(ref, c)


@pytest.mark.parametrize("name", ["text", "table", "float_csv", "float32", "int32", "lowcard", "zeros", "noise"])
def test_decode_classes(backend, lz_path, oracle, name):
    size = 3 * 65536 + 4321 if backend.name == "gpu" else 65536 + 321
    chunks = datasets.split_chunks(datasets.CLASSES[name](size, 2))
    check_decode(backend, oracle, chunks, cpu_compress(oracle, chunks))


def test_batch_that_fills_the_card(backend, oracle):
    """From 8 192 chunks on the window decoder runs in one-wave workgroups (api/snappy_api.hip)."""
    if backend.name != "gpu":
        pytest.skip("a launch shape of the GPU: 8 200 workgroups take the emulator half a minute")
    data = datasets.silesia_style(8200 * 384, 4)
    chunks = datasets.split_chunks(data, 384)
    assert len(chunks) >= 8192
    check_decode(backend, oracle, chunks, cpu_compress(oracle, chunks))


def test_reference_synth_workload(backend, lz_path, oracle):
    """benchmark_snappy_synth: 64 KiB chunks of uniform bytes in [0,3], the same device array
    passed as capacity and as actual-size output (benchmarks/benchmark_snappy_synth.cpp:244-245)."""
    n = 8 if backend.name == "gpu" else 2
    chunks = datasets.split_chunks(datasets.gen_data(3, n * 65536, 0))
    comp = cpu_compress(oracle, chunks)
    from nvcomp_amd.batched import empty_batch, make_batch, read_batch

    d = backend.dev
    codec = backend.codec("Snappy")
    cb = make_batch(d, comp, align=1)
    ob = empty_batch(d, [65536] * n, stride=65536)
    statuses = d.upload(np.full(n, -1, dtype=np.int32).view(np.uint8))
    rc = codec.decompress_async(cb, ob, ob.sizes, statuses, None, 0)  # actual aliases capacities
    d.synchronize()
    assert rc == 0
    assert (d.download(statuses).view(np.int32)[:n] == 0).all()
    assert d.download(ob.sizes).view(np.uint64)[:n].tolist() == [65536] * n
    for o, c in zip(read_batch(d, ob), chunks):
        assert np.array_equal(o, c)


def _lit(data):
    n = len(data) - 1
    if n < 60:
        return bytes([n << 2]) + bytes(data)
    nb = (n.bit_length() + 7) // 8
    return bytes([(59 + nb) << 2]) + n.to_bytes(nb, "little") + bytes(data)


def _varint(v):
    out = b""
    while v >= 128:
        out += bytes([(v & 127) | 128])
        v >>= 7
    return out + bytes([v])


def test_every_element_kind(backend, lz_path, oracle):
    """Hand-built legal streams: copy-1, copy-2, copy-4, 1..4-byte literal lengths,
    overlapping copies with offsets 1, 2, 3, copies of length 1..3 (copy-2 form)."""
    rng = np.random.RandomState(5)
    streams, raws = [], []
    # A: literal + copy1 + copy2 + copy4 + overlaps
    lit = rng.randint(0, 256, size=100).astype(np.uint8).tobytes()
    body = _lit(lit)
    raw = bytearray(lit)

    def copy(kind, off, ln):
        nonlocal body
        if kind == 1:
            assert 4 <= ln <= 11 and off < 2048
            body += bytes([1 | ((ln - 4) << 2) | ((off >> 8) << 5), off & 255])
        elif kind == 2:
            body += bytes([2 | ((ln - 1) << 2)]) + off.to_bytes(2, "little")
        else:
            body += bytes([3 | ((ln - 1) << 2)]) + off.to_bytes(4, "little")
        for _ in range(ln):
            raw.append(raw[-off])

    copy(1, 10, 7)
    copy(2, 50, 64)
    copy(3, 100, 33)
    copy(2, 1, 64)   # run of one byte
    copy(2, 2, 63)
    copy(2, 3, 5)
    copy(2, 7, 1)
    copy(2, 9, 2)
    copy(2, 11, 3)
    copy(1, 4, 11)
    body += _lit(b"xyz")
    raw += b"xyz"
    copy(3, len(raw), 64)  # offset == everything produced so far
    streams.append(_varint(len(raw)) + body)
    raws.append(bytes(raw))
    # B: literals with 1-, 2- and 3-byte length fields
    parts, raw2 = b"", b""
    for ln in (60, 61, 255, 256, 257, 4000, 70000):
        blk = rng.randint(0, 256, size=ln).astype(np.uint8).tobytes()
        parts += _lit(blk)
        raw2 += blk
    streams.append(_varint(len(raw2)) + parts)
    raws.append(raw2)
    # C: 4-byte literal length field, written non-minimally (legal)
    blk = rng.randint(0, 256, size=300).astype(np.uint8).tobytes()
    streams.append(_varint(300) + bytes([63 << 2]) + (299).to_bytes(4, "little") + blk)
    raws.append(blk)
    # D: empty buffer = preamble only
    streams.append(_varint(0))
    raws.append(b"")
    chunks = [np.frombuffer(r, dtype=np.uint8) for r in raws]
    comp = [np.frombuffer(s, dtype=np.uint8) for s in streams]
    if oracle.have_ref():
        for s, r in zip(comp, chunks):
            rc, out = oracle.ref_snappy_decompress(s, max(r.size, 1))
            assert rc == 0 and np.array_equal(out, r), "hand-built stream is not legal snappy"
    check_decode(backend, oracle, chunks, comp)
    sizes = backend.codec("Snappy").get_decompress_size(comp)
    assert sizes.tolist() == [c.size for c in chunks]


def test_copy_trains(backend, lz_path, oracle):
    """Trains of copy elements with the same offset and no literal between them -- how the format spells a match longer
    than 64 bytes. The decoder merges a train into one match per batch (snappy_decode_window.hip.h): trains shorter and
    longer than a batch, trains longer than the window, overlapping periods 1..7, mixed copy-1/2/4 encodings of the
    same offset, trains cut by a literal or by a change of offset, and a train right at the start of a batch."""
    rng = np.random.RandomState(77)
    streams, raws = [], []

    def build(program):
        body, raw = b"", bytearray()
        for op in program:
            if op[0] == "lit":
                blk = rng.randint(0, 256, size=op[1]).astype(np.uint8).tobytes()
                body += _lit(blk)
                raw += blk
            else:
                _, kind, off, ln = op
                assert 0 < off <= len(raw)
                if kind == 1:
                    assert 4 <= ln <= 11 and off < 2048
                    body += bytes([1 | ((ln - 4) << 2) | ((off >> 8) << 5), off & 255])
                elif kind == 2:
                    body += bytes([2 | ((ln - 1) << 2)]) + off.to_bytes(2, "little")
                else:
                    body += bytes([3 | ((ln - 1) << 2)]) + off.to_bytes(4, "little")
                for _ in range(ln):
                    raw.append(raw[-off])
        streams.append(_varint(len(raw)) + body)
        raws.append(bytes(raw))

    # one long train per period, far longer than window + batch
    for off in (1, 2, 3, 4, 5, 7, 64, 100):
        build([("lit", 100)] + [("copy", 2, off, 64)] * 150)
    # trains of random length and element sizes, same offset, separated by literals or by an offset change
    for _ in range(6):
        prog = [("lit", 300)]
        produced = 300
        for _ in range(200):
            if produced > 60000:  # stay inside one 64 KiB chunk (2-byte offsets)
                break
            off = int(rng.choice([1, 3, 8, 63, 64, 65, 200, produced]))
            off = min(off, produced)
            for _ in range(int(rng.choice([1, 2, 3, 5, 17, 40, 70]))):
                kind = int(rng.choice([1, 2, 3])) if off < 2048 else int(rng.choice([2, 3]))
                ln = int(rng.randint(4, 12)) if kind == 1 else int(rng.choice([1, 2, 3, 4, 31, 32, 33, 63, 64]))
                prog.append(("copy", kind, off, ln))
                produced += ln
            if rng.rand() < 0.5:
                n = int(rng.choice([1, 2, 3, 4, 5, 60, 61]))
                prog.append(("lit", n))
                produced += n
        build(prog)
    # a train that starts as the very first element after a 1-byte literal, and 63 one-byte copies then a long one
    build([("lit", 1)] + [("copy", 2, 1, 64)] * 40 + [("lit", 2)] + [("copy", 2, 2, 1)] * 63 + [("copy", 2, 2, 64)] * 30)
    chunks = [np.frombuffer(r, dtype=np.uint8) for r in raws]
    comp = [np.frombuffer(s, dtype=np.uint8) for s in streams]
    if oracle.have_ref():
        for s, r in zip(comp, chunks):
            rc, out = oracle.ref_snappy_decompress(s, max(r.size, 1))
            assert rc == 0 and np.array_equal(out, r), "hand-built stream is not legal snappy"
    check_decode(backend, oracle, chunks, comp)
    check_decode(backend, oracle, chunks, comp, checked=False)


def test_column_runs(backend, lz_path, oracle):
    """Typed columns in Snappy's spelling: a literal element of a few bytes, then a train of copies of period 1, 2, 4, 8
    or 16: every period, literal elements of 1 .. 17 bytes, trains from 4 bytes to 2 000, mixed element encodings, a
    period that changes, ordinary elements in between, every output alignment; and the int32 / sorted-key columns through
    libsnappy. (The shape of LZ4's run batches, tests/test_lz4_decode.py::test_run_batches; Snappy's decoders leave the
    run executor off -- snappy_decode_window.hip.h -- and merge the trains instead.)"""
    rng = np.random.RandomState(909)
    few = backend.name == "emu"
    streams, raws = [], []

    def build(program):
        body, raw = b"", bytearray()
        for op in program:
            if op[0] == "lit":
                blk = rng.randint(0, 256, size=op[1]).astype(np.uint8).tobytes()
                body += _lit(blk)
                raw += blk
            else:
                _, off, total = op
                assert 0 < off <= len(raw)
                while total:
                    ln = min(total, int(rng.choice([64, 64, 64, 60, 33, 11, 4])))
                    if 4 <= ln <= 11 and rng.rand() < 0.5:
                        body += bytes([1 | ((ln - 4) << 2) | ((off >> 8) << 5), off & 255])
                    else:
                        body += bytes([2 | ((ln - 1) << 2)]) + off.to_bytes(2, "little")
                    for _ in range(ln):
                        raw.append(raw[-off])
                    total -= ln
        streams.append(_varint(len(raw)) + body)
        raws.append(bytes(raw))

    for off in (1, 2, 4, 8, 16):
        for lit_choices, run_choices in (([3], [57, 64, 200, 505, 1000]), (list(range(1, 18)), [16, 17, 31, 32, 33, 48, 100, 400, 2000]),
                                         ([1, 2], [4, 9, 15, 16, 64, 300]), ([1, 4, 16, 17, 30], [64, 128])):
            prog = [("lit", max(off, 3) + int(rng.randint(14)))]
            for _ in range(25 if few else 120):
                prog.append(("copy", off, int(rng.choice(run_choices))))
                prog.append(("lit", int(rng.choice(lit_choices))))
            build(prog)
    for k in range(2 if few else 8):
        prog = [("lit", 40)]
        for j in range(60 if few else 250):
            kind = rng.randint(10)
            off = (1, 2, 4, 8, 16)[(j // 11 + k) % 5]
            if kind < 7:
                prog += [("lit", int(rng.randint(1, 5))), ("copy", off, 16 + int(rng.randint(500)))]
            elif kind < 9:
                prog += [("lit", int(rng.randint(1, 20))), ("copy", 1 + int(rng.randint(30)), 4 + int(rng.randint(40)))]
            else:
                prog += [("copy", off, 4 + int(rng.randint(12)))]
        build(prog)
    chunks = [np.frombuffer(r, dtype=np.uint8) for r in raws]
    comp = [np.frombuffer(s, dtype=np.uint8) for s in streams]
    for name in ("mortgage_col0_like", "int32"):
        gen = getattr(datasets, name) if hasattr(datasets, name) else datasets.CLASSES[name]
        cs = datasets.split_chunks(gen(65536 + 4096, 3))
        chunks += cs
        comp += cpu_compress(oracle, cs)
    if oracle.have_ref():
        for s, r in zip(comp, chunks):
            rc, out = oracle.ref_snappy_decompress(s, max(r.size, 1))
            assert rc == 0 and np.array_equal(out, r), "hand-built stream is not legal snappy"
    for mis in ((0, 5) if few else range(16)):
        check_decode(backend, oracle, chunks, comp, base_misalign=mis)
    check_decode(backend, oracle, chunks, comp, checked=False)


def test_corrupt_streams(backend, lz_path, oracle):
    rng = np.random.RandomState(23)
    chunks = datasets.split_chunks(datasets.table_rows(24000, 4), 4000)
    comp = cpu_compress(oracle, chunks)
    bad, caps = [], []
    for c, raw in zip(comp, chunks):
        b = c.copy()
        kind = rng.randint(0, 5)
        if kind == 0:
            b = b[: rng.randint(1, b.size)]
        elif kind == 1:
            b[rng.randint(0, b.size)] ^= 1 << rng.randint(0, 8)
        elif kind == 2:
            b = np.concatenate([b, rng.randint(0, 256, size=3).astype(np.uint8)])
        elif kind == 3:
            b[0] ^= 1  # preamble disagrees with the elements
        bad.append(b)
        caps.append(raw.size if kind != 4 else raw.size - 1)
    outs, actual, status = backend.codec("Snappy").decompress(bad, caps)
    for i, (b, cap) in enumerate(zip(bad, caps)):
        rc, ref = oracle.snappy_decompress(b, cap)
        if rc == 0:
            assert status[i] == NvcompStatus.Success and actual[i] == ref.size
            assert np.array_equal(outs[i][: ref.size], ref)
        else:
            assert status[i] != NvcompStatus.Success and actual[i] == 0


@pytest.mark.parametrize("name", ["text", "table", "float_csv", "float32", "int32", "lowcard", "zeros", "noise"])
def test_compress_decodes_on_cpu(backend, oracle, name):
    size = 2 * 65536 + 77 if backend.name == "gpu" else 65536 + 77
    chunks = datasets.split_chunks(datasets.CLASSES[name](size, 6))
    codec = backend.codec("Snappy")
    comp = codec.compress(chunks)
    bound = codec.max_compressed_size(65536)
    assert bound == 32 + 65536 + 65536 // 6
    for cc, c in zip(comp, chunks):
        assert cc.size <= bound
        rc, out = oracle.snappy_decompress(cc, c.size)
        assert rc == 0 and np.array_equal(out, c)
        if oracle.have_ref():
            rc, out = oracle.ref_snappy_decompress(cc, c.size)  # snappy::RawUncompress
            assert rc == 0 and np.array_equal(out, c)
    ours = sum(c.size for c in comp)
    cpu = sum(oracle.snappy_compress(c).size for c in chunks)
    assert ours <= cpu * 1.35 + 64, (ours, cpu)


def test_roundtrip_ragged(backend, lz_path, oracle):
    rng = np.random.RandomState(9)
    base = datasets.text(9000, 3)
    sizes = [0, 1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 59, 60, 61, 62, 63, 64, 65, 66, 67, 68, 69, 127, 128, 1000, 8000]
    chunks = [base[rng.randint(0, 500):][:s].copy() for s in sizes]
    codec = backend.codec("Snappy")
    comp = codec.compress(chunks, in_align=1)
    outs, actual, status = codec.decompress(comp, [max(c.size, 0) for c in chunks])
    assert (status == 0).all() and actual.tolist() == [c.size for c in chunks]
    for o, c in zip(outs, chunks):
        assert np.array_equal(o, c)


def _varint(n):
    out = bytearray()
    while n >= 128:
        out.append((n & 127) | 128)
        n >>= 7
    out.append(n)
    return bytes(out)


def test_stream_that_grows_behind_a_compressible_front(backend, lz_path, oracle):
    """A legal stream libsnappy never writes: 40 KiB of one byte as 64-byte copy elements (3 stream bytes each), then
    25.5 KiB of one-byte literal elements (2 stream bytes each). Its total size is below a chunk's, but behind the run
    the stream is TWICE the output that is left -- the workgroup-per-chunk decoder keeps output and stream in one LDS
    buffer ("in place", common/lz_team.hip.h) and must notice that this chunk's output would overrun its unread stream
    and hand it to the one-wave decoder; every other path just decodes it. Plus the same shape with the expanding part in
    front (no conflict) and a chunk of nothing but one-byte literals (larger than a team takes: refused by size)."""
    rng = np.random.default_rng(5)

    def build(front_run, lits):
        body = bytearray()
        out = bytearray()
        if front_run:
            body += bytes([0 << 2 | 0, 0x41])  # literal "A"
            out += b"A"
            while len(out) < front_run:
                n = min(64, front_run - len(out))
                body += bytes([((n - 1) << 2) | 2, 1, 0])  # copy-2: length n, offset 1
                out += b"A" * n
        for v in lits:
            body += bytes([0, v])  # literal element of one byte
            out.append(v)
        return out, body

    cases = []
    out, body = build(40 * 1024, rng.integers(0, 256, 65536 - 40 * 1024, dtype=np.uint8).tolist())
    cases.append((out, body))
    lits = rng.integers(0, 256, 12 * 1024, dtype=np.uint8).tolist()
    o2, b2 = build(0, lits)
    tail_out = bytearray(o2)
    tail_body = bytearray(b2)
    while len(tail_out) < 65536:  # expanding part first, the run behind it
        n = min(64, 65536 - len(tail_out))
        tail_body += bytes([((n - 1) << 2) | 2, 1, 0])
        tail_out += bytes([tail_out[-1]]) * n
    cases.append((tail_out, tail_body))
    cases.append(build(0, rng.integers(0, 256, 40000, dtype=np.uint8).tolist()))
    chunks = [np.frombuffer(bytes(o), dtype=np.uint8) for o, _ in cases]
    comp = [np.frombuffer(_varint(len(o)) + bytes(b), dtype=np.uint8) for o, b in cases]
    assert comp[0].size < 65536 + 512  # small enough for a team to take it
    for c, cc in zip(chunks, comp):
        if oracle.have_ref():
            rc, ref = oracle.ref_snappy_decompress(cc, c.size)
            assert rc == 0 and np.array_equal(ref, c)  # libsnappy reads it
    check_decode(backend, oracle, chunks, comp)


@pytest.mark.parametrize("fmt", ["Snappy", "LZ4"])
def test_compress_small_chunk_at_the_end_of_a_mapping(emu, oracle, fmt):
    """the wide compressor's 12-byte word-check load ran past a chunk of 8 .. 11 bytes (Snappy looks for
    matches from 8 bytes on). On the emulator the kernels are host code: chunks of 1 .. 16 bytes that END on the last byte in
    front of a PROT_NONE page are compressed in place -- a read past the end is a segfault."""
    import ctypes as C
    import mmap

    libc = C.CDLL(None, use_errno=True)
    libc.mmap.restype = C.c_void_p
    libc.mmap.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
    libc.mprotect.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    page = mmap.PAGESIZE
    base = libc.mmap(None, 2 * page, mmap.PROT_READ | mmap.PROT_WRITE, mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS, -1, 0)
    assert base not in (None, C.c_void_p(-1).value)
    assert libc.mprotect(base + page, page, 0) == 0  # PROT_NONE
    codec = emu.codec(fmt)
    lib = emu.lib
    sizes = list(range(1, 17))
    for n in sizes:
        raw = (np.arange(n, dtype=np.uint8) % 3 + 65).astype(np.uint8)  # compressible: "ABCABC..."
        C.memmove(base + page - n, raw.ctypes.data, n)
        in_ptrs = np.array([base + page - n], dtype=np.uint64)
        in_sizes = np.array([n], dtype=np.uint64)
        max_out = codec.max_compressed_size(64)
        out = np.zeros(max_out, dtype=np.uint8)
        out_ptrs = np.array([out.ctypes.data], dtype=np.uint64)
        out_sizes = np.zeros(1, dtype=np.uint64)
        tb = codec.compress_temp_size(1, 64)
        temp = np.zeros(max(tb, 1), dtype=np.uint8)
        fn = getattr(lib, f"nvcompBatched{fmt}CompressAsync")
        rc = fn(in_ptrs.ctypes.data, in_sizes.ctypes.data, 64, 1, temp.ctypes.data, tb, out_ptrs.ctypes.data,
                out_sizes.ctypes.data, codec.opts, None)
        assert rc == 0
        dec = oracle.ref_snappy_decompress if fmt == "Snappy" else oracle.ref_lz4_decompress
        code, back = dec(out[: int(out_sizes[0])], n)
        assert code == 0 and np.array_equal(back, raw), (fmt, n)


@pytest.mark.parametrize("fmt", ["Snappy", "LZ4"])
def test_compress_runs_at_the_end_of_a_mapping(emu, oracle, fmt):
    """The same for the run compressor (common/lz_match_runs.hip.h: 16-byte lane loads of the chunk and of the bytes in front of
    it): chunks of runs whose last byte is the last one in front of a PROT_NONE page, sizes around the 1 KiB steps and the
    16-byte lanes."""
    import ctypes as C
    import mmap

    libc = C.CDLL(None, use_errno=True)
    libc.mmap.restype = C.c_void_p
    libc.mmap.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
    libc.mprotect.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    page = mmap.PAGESIZE
    span = 20 * page
    base = libc.mmap(None, span + page, mmap.PROT_READ | mmap.PROT_WRITE, mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS, -1, 0)
    assert base not in (None, C.c_void_p(-1).value)
    assert libc.mprotect(base + span, page, 0) == 0  # PROT_NONE behind the chunk
    codec = emu.codec(fmt)
    lib = emu.lib
    for n in (4096, 4097, 4111, 4112, 4113, 5119, 5120, 5121, 8191, 65535, 65536):
        raw = np.repeat((np.arange(n // 8 + 1) // 37).astype(np.uint64) + np.uint64(10 ** 11), 1).view(np.uint8)[:n].copy()
        C.memmove(base + span - n, raw.ctypes.data, n)
        in_ptrs = np.array([base + span - n], dtype=np.uint64)
        in_sizes = np.array([n], dtype=np.uint64)
        max_out = codec.max_compressed_size(65536)
        out = np.zeros(max_out, dtype=np.uint8)
        out_ptrs = np.array([out.ctypes.data], dtype=np.uint64)
        out_sizes = np.zeros(1, dtype=np.uint64)
        tb = codec.compress_temp_size(1, 65536)
        temp = np.zeros(max(tb, 1), dtype=np.uint8)
        fn = getattr(lib, f"nvcompBatched{fmt}CompressAsync")
        rc = fn(in_ptrs.ctypes.data, in_sizes.ctypes.data, 65536, 1, temp.ctypes.data, tb, out_ptrs.ctypes.data,
                out_sizes.ctypes.data, codec.opts, None)
        assert rc == 0
        dec = oracle.ref_snappy_decompress if fmt == "Snappy" else oracle.ref_lz4_decompress
        code, back = dec(out[: int(out_sizes[0])], n)
        assert code == 0 and np.array_equal(back, raw), (fmt, n)
        assert int(out_sizes[0]) * 8 < n, "the run compressor's ratio"

