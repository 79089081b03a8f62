"""LZ4 batched compress parity. The reference pins the compressor only through
decodability by liblz4 (examples/lz4_cpu_decompression.cu:142-157) and the round
trip of benchmarks/benchmark_template_chunked.cuh:553-584; both are checked, plus
the declared worst-case bound and a ratio sanity check against the CPU "fast" class."""
import numpy as np
import pytest

from nvcomp_amd import datasets


def decode_all(oracle, comp, chunks):
    for i, (cc, c) in enumerate(zip(comp, chunks)):
        rc, out = oracle.lz4_decompress(cc, c.size)
        assert rc == 0 and np.array_equal(out, c), f"oracle rejects chunk {i}"
        if oracle.have_ref():
            rc, out = oracle.ref_lz4_decompress(cc, c.size)  # LZ4_decompress_safe
            assert rc == 0 and np.array_equal(out, c), f"liblz4 rejects chunk {i}"


@pytest.mark.parametrize("name", ["text", "table", "float_csv", "float32", "int32", "lowcard", "zeros", "noise"])
def test_compress_decodes_on_cpu(backend, oracle, name):
    size = 2 * 65536 + 999 if backend.name == "gpu" else 65536 + 999
    chunks = datasets.split_chunks(datasets.CLASSES[name](size, 4))
    codec = backend.codec("LZ4")
    comp = codec.compress(chunks)
    decode_all(oracle, comp, chunks)
    bound = codec.max_compressed_size(65536)
    assert bound == 65536 + 65536 // 255 + 16
    assert all(c.size <= bound for c in comp)
    ours = sum(c.size for c in comp)
    cpu = sum(oracle.lz4_compress(c).size for c in chunks)
    assert ours <= cpu * 1.35 + 64, (ours, cpu)


def test_compress_tiny_and_ragged(backend, oracle):
    rng = np.random.RandomState(3)
    base = datasets.table_rows(9000, 5)
    sizes = [0, 1, 2, 4, 5, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 63, 64, 65, 66, 127, 128, 129, 300, 4096, 8191]
    chunks = [base[rng.randint(0, 500):][:s].copy() for s in sizes]
    comp = backend.codec("LZ4").compress(chunks, in_align=1)
    decode_all(oracle, comp, chunks)


def test_gpu_roundtrip(backend, oracle):
    chunks = datasets.split_chunks(datasets.silesia_style(6 * 16384, 8, chunk=16384), 16384)
    codec = backend.codec("LZ4")
    comp = codec.compress(chunks)
    outs, actual, status = codec.decompress(comp, [c.size for c in chunks])
    assert (status == 0).all() and actual.tolist() == [c.size for c in chunks]
    for o, c in zip(outs, chunks):
        assert np.array_equal(o, c)


def test_opts_validation(backend):
    import ctypes as C

    from nvcomp_amd._lib import LZ4Opts, NvcompStatus

    lib = backend.lib
    out = C.c_size_t(0)
    assert lib.nvcompBatchedLZ4CompressGetMaxOutputChunkSize(65536, LZ4Opts(7), C.byref(out)) == NvcompStatus.ErrorInvalidValue
    assert lib.nvcompBatchedLZ4CompressGetMaxOutputChunkSize(65536, LZ4Opts(0xFF), C.byref(out)) == NvcompStatus.Success
    assert lib.nvcompBatchedLZ4CompressGetTempSize(10, 1 << 25, LZ4Opts(0), C.byref(out)) == NvcompStatus.ErrorChunkSizeTooLarge


@pytest.mark.parametrize("data_type,width", [(2, 2), (3, 2), (4, 4), (5, 4)])
def test_data_type_option_is_honoured(backend, oracle, data_type, width):
    """nvcompBatchedLZ4Opts_t.data_type (benchmarks/benchmark_lz4_chunked.cu:32,43,76-84; CHANGELOG.md:168-169): with a
    2- or 4-byte element type matches are searched at element boundaries only. Every block still decodes with
    LZ4_decompress_safe, whatever the chunk length (a ragged tail is literals), and on the int32 column (liblz4
    ratio 37.5) the typed search loses nothing."""
    chunks = datasets.split_chunks(datasets.int32_column(2 * 65536, 5)) + [datasets.text(30000, 2),
                                                                          datasets.float32_column(65536, 1),
                                                                          datasets.int32_column(4 * 1000 + 3, 9)]
    typed = backend.codec("LZ4", (data_type,)).compress(chunks)
    decode_all(oracle, typed, chunks)
    untyped = backend.codec("LZ4").compress(chunks)
    col = slice(0, 2)  # the two whole int32-column chunks
    r_typed = sum(c.size for c in chunks[col]) / sum(c.size for c in typed[col])
    r_untyped = sum(c.size for c in chunks[col]) / sum(c.size for c in untyped[col])
    cpu = sum(c.size for c in chunks[col]) / sum(oracle.lz4_compress(c).size for c in chunks[col])
    assert r_untyped >= 33 and r_untyped >= 0.9 * cpu, (r_untyped, cpu)
    if width == 4:
        assert r_typed >= 0.9 * r_untyped, (r_typed, r_untyped)


def test_oversized_chunk_is_not_compressed(backend):
    """A chunk larger than max_uncompressed_chunk_bytes would overrun the slot sized from GetMaxOutputChunkSize: its
    compressed size reads 0 and nothing is written."""
    from nvcomp_amd.batched import BatchedCodec

    chunks = [datasets.text(3000, 1), datasets.text(9000, 2), datasets.text(2000, 3)]
    codec = backend.codec("LZ4")
    comp = codec.compress(chunks, max_chunk=4096) if "max_chunk" in BatchedCodec.compress.__code__.co_varnames else None
    if comp is None:
        pytest.skip("harness cannot declare a smaller max chunk")
    assert comp[1].size == 0 and comp[0].size > 0 and comp[2].size > 0


@pytest.mark.parametrize("fmt,opts", [("Snappy", None), ("ANS", None), ("Bitcomp", (0, 1)), ("Cascaded", (4096, 4, 2, 1, 1))])
def test_oversized_chunk_is_refused_by_every_compressor(backend, fmt, opts):
    """Same rule for every format: a chunk larger than the declared max_uncompressed_chunk_bytes comes back with size 0."""
    chunks = [datasets.int32_column(3000, 1), datasets.int32_column(9000, 2), datasets.int32_column(2000, 3)]
    comp = backend.codec(fmt, opts).compress(chunks, max_chunk=4096)
    assert comp[1].size == 0 and comp[0].size > 0 and comp[2].size > 0


@pytest.mark.parametrize("fmt", ["LZ4", "Snappy"])
def test_compress_fuzz_structured(backend, oracle, fmt):
    """Chunks built to hit the corners of the 256-position steps (common/lz_match_wide.hip.h): runs and periods that start
    and end at every alignment of the step and of the image blocks, matches that continue at the same distance behind a
    few changed bytes, long matches followed by text, sizes around the multiples of 256, unaligned chunk starts. Every
    stream must decode with the CPU library to the original."""
    rng = np.random.RandomState(77)
    text = datasets.text(1 << 16, 9)
    chunks = []
    for i in range(36 if backend.name == "gpu" else 14):
        n = int(rng.choice([255, 256, 257, 511, 513, 1023, 1280, 4095, 4097, 20000, 65535, 65536]))
        kind = i % 7
        if kind == 0:  # periodic column with occasional changed bytes (the same-distance continuation)
            period = int(rng.choice([1, 2, 3, 4, 8, 12, 16, 24]))
            c = np.tile(rng.randint(0, 256, period).astype(np.uint8), n // period + 1)[:n].copy()
            for p in rng.randint(0, n, size=max(1, n // 300)):
                c[p] ^= 1 + rng.randint(0, 255)
        elif kind == 1:  # text with a long run spliced in at a random place
            c = text[rng.randint(0, 1000):][:n].copy()
            a = rng.randint(0, max(1, n - 1))
            c[a: a + rng.randint(1, 3000)] = rng.randint(0, 256)
        elif kind == 2:  # a block repeated at a distance just below / above 64 KiB of reach and 256 of a step
            c = rng.randint(0, 256, n).astype(np.uint8)
            d = int(rng.choice([1, 255, 256, 257, 1024, 4096]))
            if n > 2 * d + 8:
                c[d: 2 * d] = c[:d]
        elif kind == 3:
            c = np.zeros(n, np.uint8)
            c[rng.randint(0, n)] = 1
        elif kind == 4:
            c = datasets.int32_column(n, i)
        elif kind == 5:
            c = np.concatenate([text[:n // 2], text[:n - n // 2]])  # the second half is one long match ... of text
        else:
            c = datasets.lowcard(n, i)
        chunks.append(np.ascontiguousarray(c[:n]))
    codec = backend.codec(fmt)
    comp = codec.compress(chunks, in_align=1)
    for i, (cc, c) in enumerate(zip(comp, chunks)):
        if fmt == "LZ4":
            rc, out = (oracle.ref_lz4_decompress if oracle.have_ref() else oracle.lz4_decompress)(cc, c.size)
        else:
            rc, out = (oracle.ref_snappy_decompress if oracle.have_ref() else oracle.snappy_decompress)(cc, c.size)
        assert rc == 0 and np.array_equal(out, c), f"chunk {i} ({c.size} bytes, kind {i % 7})"
    assert all(cc.size <= codec.max_compressed_size(max(c.size, 1)) for cc, c in zip(comp, chunks))


@pytest.mark.parametrize("fmt", ["LZ4", "Snappy"])
def test_compress_runs(backend, oracle, fmt):
    """Chunks that are their own copy from 1, 2, 4 or 8 bytes back -- sorted keys, typed columns, zeros -- take the run
    compressor (common/lz_match_runs.hip.h: 1 KiB a step, only the mismatches cost work): every period, sizes around its 4 KiB
    floor and around the 1 KiB steps and 16-byte lanes, changes at every position of a step and right at the chunk's end
    (the formats' end-of-block rules), stretches of one to three equal bytes (literals, not matches), dense changes, a first
    KiB of runs in front of noise or text (given up: the match finder starts over), unaligned chunk starts. Every stream
    decodes with the CPU library to the original; the columns keep the ratio the match finder had on them."""
    rng = np.random.RandomState(4321)
    text = datasets.text(1 << 16, 9)
    chunks = []
    sizes = [4095, 4096, 4097, 4111, 4112, 5119, 5120, 5121, 6000, 8191, 8192, 20000, 65535, 65536]
    for period in (1, 2, 4, 8):
        for n in (sizes if backend.name == "gpu" else sizes[1::3] + [4096, 65536]):
            # values that change every now and then, in their low bytes
            nvals = n // period + 2
            runs = rng.randint(1, 120, size=nvals)
            vals = np.cumsum(rng.randint(1, 300, size=nvals)).astype(np.uint64) + np.uint64(0x0102030405060708)
            col = np.repeat(vals, runs)[: n // period + 1]
            c = col.astype({1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[period]).view(np.uint8)[:n].copy()
            chunks.append(c)
    n = 65536
    base = np.repeat((np.arange(n // 8 + 1) // 50).astype(np.uint64) + np.uint64(10 ** 11), 1).view(np.uint8)[:n].copy()
    for k in range(12 if backend.name == "gpu" else 5):
        c = base.copy()
        kind = k % 6
        if kind == 0:  # single changed bytes at every position class of a step
            for p in rng.randint(16, n, size=200):
                c[p] ^= 0x5a
        elif kind == 1:  # changes right at the end of the chunk
            for p in (n - 1, n - 4, n - 5, n - 6, n - 12, n - 13, n - 17):
                c[p] ^= 0x33
        elif kind == 2:  # stretches of one to three equal bytes between changes
            for p in range(3000, 3400, 1 + k % 4 + 1):
                c[p] ^= 0x11
        elif kind == 3:  # the first KiB says runs, the rest is noise
            c[2048:] = rng.randint(0, 256, size=n - 2048)
        elif kind == 4:  # ... or text
            c[1500:] = text[: n - 1500]
        else:  # dense changes everywhere behind the first KiB
            idx = np.arange(1100, n, 5)
            c[idx] ^= 0x77
        chunks.append(c)
    chunks.append(np.zeros(65536, np.uint8))
    chunks.append(np.full(4096, 7, np.uint8))
    codec = backend.codec(fmt)
    for in_align in (16, 1):
        comp = codec.compress(chunks, in_align=in_align)
        for i, (cc, c) in enumerate(zip(comp, chunks)):
            if fmt == "LZ4":
                rc, out = (oracle.ref_lz4_decompress if oracle.have_ref() else oracle.lz4_decompress)(cc, c.size)
            else:
                rc, out = (oracle.ref_snappy_decompress if oracle.have_ref() else oracle.snappy_decompress)(cc, c.size)
            assert rc == 0 and np.array_equal(out, c), f"chunk {i} ({c.size} bytes)"
        assert all(cc.size <= codec.max_compressed_size(max(c.size, 1)) for cc, c in zip(comp, chunks))
    # the columns: what the match finder made of them in round 5 (LZ4: sorted keys 56.7, int32 37.5)
    for gen, floor in ((datasets.mortgage_col0_like, 50.0 if fmt == "LZ4" else 15.0), (datasets.int32_column, 33.0 if fmt == "LZ4" else 13.0)):
        cs = datasets.split_chunks(gen(2 * 65536, 3))
        comp = codec.compress(cs)
        ratio = sum(c.size for c in cs) / sum(cc.size for cc in comp)
        assert ratio >= floor, (gen.__name__, ratio)


@pytest.mark.parametrize("fmt", ["LZ4", "Snappy"])
def test_compress_after_a_quiet_stretch(backend, oracle, fmt):
    """Steps without a single hit switch the match finder to every fourth position (NVCOMP_LZMW_QUIET_STEPS,
    common/lz_match_wide.hip.h). What follows such a stretch must still be found: a repeat of an earlier kilobyte of the
    noise (found up to three positions late, grown backwards), a run, text -- at every alignment of the repeat's start --
    and the streams must decode with the CPU library."""
    rng = np.random.RandomState(5)
    text = datasets.text(8192, 3)
    chunks = []
    for shift in range(8):
        noise = rng.randint(0, 256, 6000 + shift).astype(np.uint8)
        chunks.append(np.concatenate([noise, noise[1000:2024], np.zeros(700, np.uint8), noise[:3], text]))
    chunks.append(rng.randint(0, 256, 65536).astype(np.uint8))
    codec = backend.codec(fmt)
    comp = codec.compress(chunks, in_align=1)
    for i, (cc, c) in enumerate(zip(comp, chunks)):
        if fmt == "LZ4":
            rc, out = (oracle.ref_lz4_decompress if oracle.have_ref() else oracle.lz4_decompress)(cc, c.size)
        else:
            rc, out = (oracle.ref_snappy_decompress if oracle.have_ref() else oracle.snappy_decompress)(cc, c.size)
        assert rc == 0 and np.array_equal(out, c), (fmt, i, c.size)
    for cc, c in zip(comp[:8], chunks[:8]):
        # the kilobyte repeated out of the noise costs a few bytes, the run a few, the text at most what it costs alone
        alone = codec.compress([text], in_align=1)[0].size
        assert cc.size <= (c.size - 1024 - 700 - text.size) + alone + 160, (fmt, cc.size, c.size, alone)
    outs, _, status = codec.decompress(comp, [c.size for c in chunks])
    assert (status == 0).all() and all(np.array_equal(o, c) for o, c in zip(outs, chunks))


@pytest.mark.parametrize("fmt", ["LZ4", "Snappy"])
def test_compress_many_measured_hits_a_step(backend, oracle, fmt):
    """More than 64 hits a step that are still open after eight bytes -- the first 64 are measured around the next step's
    probe, the others in the loop behind it (common/lz_match_wide.hip.h) -- with fewer than 60 hits in the step's first 64
    positions (no whole-wave path): 192 bytes copied from further up with one byte in twelve changed, behind 64 fresh
    bytes, step after step; also as the chunk's LAST step (no probe follows) and with the chunk ending inside the copy
    (candidates whose 24 bytes reach past the end are read dword by dword)."""
    rng = np.random.RandomState(11)
    chunks = []
    for tail in (0, 256, 256 + 100, 256 + 191, 256 + 250, 3 * 256 + 7):
        base = rng.randint(0, 256, 1024).astype(np.uint8)
        parts = [base]
        steps = 1 + tail // 256 + 2
        for sidx in range(steps):
            fresh = rng.randint(0, 256, 64).astype(np.uint8)
            a = 64 + 60 * sidx
            copy = base[a: a + 192].copy()
            copy[(3 + sidx) % 12::12] ^= 0x11 * (1 + sidx % 13)  # (its own places: no long match with an earlier copy)
            parts += [fresh, copy]
        c = np.concatenate(parts)
        chunks.append(np.ascontiguousarray(c[: 1024 + 256 * (steps - 1) + (tail % 256 if tail % 256 else 256)]))
    codec = backend.codec(fmt)
    comp = codec.compress(chunks, in_align=1)
    dec = ((oracle.ref_lz4_decompress if fmt == "LZ4" else oracle.ref_snappy_decompress) if oracle.have_ref()
           else (oracle.lz4_decompress if fmt == "LZ4" else oracle.snappy_decompress))
    for i, (cc, c) in enumerate(zip(comp, chunks)):
        rc, out = dec(cc, c.size)
        assert rc == 0 and np.array_equal(out, c), (fmt, i, c.size)
        # the copied parts cost a few bytes per eleven: well under half of the chunk behind its first KiB
        assert cc.size < 1024 + 16 + (c.size - 1024) * 0.62, (fmt, i, cc.size, c.size)
    outs, _, status = codec.decompress(comp, [c.size for c in chunks])
    assert (status == 0).all() and all(np.array_equal(o, c) for o, c in zip(outs, chunks))


@pytest.mark.parametrize("fmt", ["LZ4", "Snappy"])
def test_compress_chunks_beyond_64k(backend, oracle, fmt):
    """Chunks larger than the 64 KiB the tables' two-byte positions cover (up to nvcomp*CompressionMaxAllowedChunkSize =
    16 MiB are legal): candidates are rebuilt modulo 65 536 and must stay within the formats' 65 535-byte reach; matches
    across the 64 KiB marks, repeats at a distance of exactly 65 536 (not encodable: must not be taken), long runs."""
    rng = np.random.RandomState(5)
    sizes = [65537, 70000, 131072 + 3, 200000] + ([1 << 20, (1 << 22) + 17] if backend.name == "gpu" else [])
    text = datasets.text(1 << 16, 3)
    chunks = []
    for n in sizes:
        period = np.tile(rng.randint(0, 256, 65536).astype(np.uint8), n // 65536 + 2)[:n]  # repeats exactly 65 536 back
        mixed = np.concatenate([np.tile(text, n // text.size + 1)[: n // 2], np.zeros(n // 4, np.uint8),
                                datasets.int32_column(n - n // 2 - n // 4, 2)])
        chunks += [period.copy(), mixed[:n].copy()]
    codec = backend.codec(fmt)
    comp = codec.compress(chunks, in_align=1)
    dec = ((oracle.ref_lz4_decompress if fmt == "LZ4" else oracle.ref_snappy_decompress) if oracle.have_ref()
           else (oracle.lz4_decompress if fmt == "LZ4" else oracle.snappy_decompress))
    for i, (cc, c) in enumerate(zip(comp, chunks)):
        rc, out = dec(cc, c.size)
        assert rc == 0 and np.array_equal(out, c), f"chunk {i} ({c.size} bytes)"
    # the text half repeats at 65 536 bytes' distance too, the zeros and the column compress well: well below half
    assert sum(cc.size for cc in comp[1::2]) < 0.5 * sum(c.size for c in chunks[1::2])
