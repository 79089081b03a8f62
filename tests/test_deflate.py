"""DEFLATE / gzip batched decoder parity: the HIP path (or its host emulation) against zlib, the
CPU peer the reference's examples use -- examples/deflate_cpu_compression.cu:58-104 writes the chunks with libdeflate,
compress2 (wrapper cut off) or deflateInit2(-15), examples/gzip_gpu_decompression.cu:57-81 with deflateInit2(15 | 16),
and both check the GPU's output against the original bytes. Bit-exact: every byte, every size, every status."""
import gzip
import io
import zlib

import numpy as np
import pytest

from nvcomp_amd import datasets
from nvcomp_amd._lib import NvcompStatus


def raw_deflate(chunk, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0):
    """What examples/deflate_cpu_compression.cu:82-104 (algo 2) produces: deflateInit2(level, Z_DEFLATED, -15, 8, strategy)."""
    o = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    data = chunk.tobytes()
    out = b""
    if flush_every:
        for i in range(0, len(data), flush_every):  # Z_FULL_FLUSH: an empty stored block + a fresh block each time
            out += o.compress(data[i: i + flush_every]) + o.flush(zlib.Z_FULL_FLUSH)
    else:
        out += o.compress(data)
    return np.frombuffer(out + o.flush(), dtype=np.uint8)


def compress2_without_wrapper(chunk):
    """examples/deflate_cpu_compression.cu:69-81 (algo 1): compress2(level 9), then the 2-byte zlib header and the
    4-byte Adler-32 are cut off."""
    z = zlib.compress(chunk.tobytes(), 9)
    return np.frombuffer(z[2:-4], dtype=np.uint8)


def gzip_member(chunk, level=9):
    """examples/gzip_gpu_decompression.cu:57-81: deflateInit2(9, Z_DEFLATED, 15 | 16, 8, Z_DEFAULT_STRATEGY)."""
    o = zlib.compressobj(level, zlib.DEFLATED, 15 | 16)
    return np.frombuffer(o.compress(chunk.tobytes()) + o.flush(), dtype=np.uint8)


def check(backend, fmt, chunks, comp, **kw):
    codec = backend.codec(fmt)
    caps = [max(c.size, 1) for c in chunks]
    outs, actual, status = codec.decompress(comp, caps, **kw)
    if status is not None:
        assert (status == NvcompStatus.Success).all(), status
    if actual is not None:
        assert actual.tolist() == [c.size for c in chunks]
    for i, (o, c, cc) in enumerate(zip(outs, chunks, comp)):
        assert np.array_equal(o[: c.size], c), f"chunk {i} differs"
        wbits = -15 if fmt == "Deflate" else 15 | 16
        assert zlib.decompress(cc.tobytes(), wbits) == c.tobytes()  # the CPU peer reads the same bytes


CLASSES = ["text", "table", "float_csv", "float32", "int32", "lowcard", "zeros", "noise"]


def test_libdeflate_streams(backend, oracle):
    """The reference's DEFAULT CPU producer (examples/deflate_cpu_compression.cu:60-67, algo 0: libdeflate_alloc_compressor(6),
    libdeflate_deflate_compress) through the container's libdeflate (oracle/_ref shim): its block splitting and code
    construction differ from zlib's; every stream must decode to the original bytes, and libdeflate must read them back
    too (the checker of the checker)."""
    if not oracle.have_libdeflate():
        pytest.skip("libdeflate is not in this container")
    size = 65536 + 4321 if backend.name == "emu" else 3 * 65536 + 4321
    chunks = []
    for name in CLASSES:
        chunks += datasets.split_chunks(datasets.CLASSES[name](size, 7))
    comp = [oracle.ref_libdeflate_compress(c) for c in chunks]
    for c, cc in zip(chunks[::5], comp[::5]):
        rc, back = oracle.ref_libdeflate_decompress(cc, c.size)
        assert rc == 0 and np.array_equal(back, c)
    check(backend, "Deflate", chunks, comp)


@pytest.mark.parametrize("name", CLASSES)
def test_decode_classes(backend, name):
    size = 3 * 65536 + 4321 if backend.name == "gpu" else 65536 + 4321
    chunks = datasets.split_chunks(datasets.CLASSES[name](size, 2))
    check(backend, "Deflate", chunks, [raw_deflate(c, 9) for c in chunks])


@pytest.mark.parametrize("how", ["level1", "level6", "fixed", "huffman_only", "rle", "stored", "compress2", "many_blocks"])
def test_block_kinds(backend, how):
    """Dynamic, fixed and stored blocks, literal-only streams, several blocks per chunk (zlib starts a new block
    every 16 K symbols or at a flush; a full flush also leaves an empty stored block in the stream)."""
    size = 4 * 65536 if backend.name == "gpu" else 65536 + 999
    chunks = datasets.split_chunks(datasets.silesia_style(size, 5))
    make = {
        "level1": lambda c: raw_deflate(c, 1),
        "level6": lambda c: raw_deflate(c, 6),
        "fixed": lambda c: raw_deflate(c, 6, zlib.Z_FIXED),
        "huffman_only": lambda c: raw_deflate(c, 6, zlib.Z_HUFFMAN_ONLY),
        "rle": lambda c: raw_deflate(c, 6, zlib.Z_RLE),
        "stored": lambda c: raw_deflate(c, 0),
        "compress2": compress2_without_wrapper,
        "many_blocks": lambda c: raw_deflate(c, 6, flush_every=3000),
    }[how]
    check(backend, "Deflate", chunks, [make(c) for c in chunks])


def test_many_small_streams(backend):
    """A few hundred chunks of every size and kind in one batch, all levels and strategies: block boundaries, window
    boundaries of the speculative decoder and batch cuts fall everywhere."""
    rng = np.random.RandomState(17)
    pool = {n: datasets.CLASSES[n](1 << 18, 4) for n in ("text", "table", "float32", "lowcard", "noise", "zeros", "int32")}
    chunks, comp = [], []
    for i in range(240 if backend.name == "gpu" else 40):
        name = list(pool)[i % len(pool)]
        size = int(rng.randint(1, 20000))
        at = int(rng.randint(0, (1 << 18) - size))
        c = pool[name][at: at + size].copy()
        if i % 5 == 0:  # splice two kinds
            c[size // 2:] = pool["text"][: size - size // 2]
        level = int(rng.randint(1, 10))
        strategy = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_RLE, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY][i % 5 if i % 3 == 0 else 0]
        chunks.append(c)
        comp.append(raw_deflate(c, level, strategy, flush_every=int(rng.randint(200, 4000)) if i % 7 == 0 else 0))
    check(backend, "Deflate", chunks, comp)


def test_ragged_and_tiny_chunks(backend):
    rng = np.random.RandomState(3)
    text = datasets.CLASSES["text"](40000, 1)
    chunks = [np.zeros(0, np.uint8), np.frombuffer(b"a", np.uint8), np.frombuffer(b"ab" * 5, np.uint8), text[:13], text[:4097],
              text[:31999], rng.randint(0, 256, 70, dtype=np.uint8), np.zeros(65536, np.uint8)]
    for level in (9, 0):
        check(backend, "Deflate", chunks, [raw_deflate(c, level) for c in chunks], comp_align=1, out_align=1, base_misalign=3)


class BitWriter:
    """LSB-first bit packing of RFC 1951; Huffman codes go in most significant bit first."""

    def __init__(self):
        self.bits = []

    def put(self, value, n, msb_first=False):
        for i in range(n):
            self.bits.append((value >> (n - 1 - i if msb_first else i)) & 1)

    def align(self):
        while len(self.bits) % 8:
            self.bits.append(0)

    def raw(self, data):
        assert len(self.bits) % 8 == 0
        for byte in data:
            self.put(byte, 8)

    def bytes(self):
        out = bytearray((len(self.bits) + 7) // 8)
        for i, b in enumerate(self.bits):
            out[i // 8] |= b << (i % 8)
        return bytes(out)


def test_longest_matches_and_farthest_distances(backend):
    """Length 258 at distance 1 (a run), matches whose source lies ~32 K back (the executor reads those from HBM, the
    window holds 2 KiB), and -- assembled by hand, zlib never emits it -- length 258 at the format's largest
    distance, 32 768."""
    rng = np.random.RandomState(11)
    block = rng.randint(0, 256, 32500, dtype=np.uint8)
    far = np.concatenate([block, block])                       # every match 32 500 back
    near_far = np.concatenate([block[:20000], rng.randint(0, 256, 12500, dtype=np.uint8), block[:20000]])
    run = np.full(65536, 7, np.uint8)
    chunks = [far, near_far, run]
    comp = [raw_deflate(c, 9) for c in chunks]
    assert comp[0].size < 40000  # zlib did find the far matches
    # stored block of 32 768 bytes, then a fixed block: <length 258, distance 32 768> x 3, 'Z', end of block
    head = rng.randint(0, 256, 32768, dtype=np.uint8)
    w = BitWriter()
    w.put(0, 1), w.put(0, 2), w.align()
    w.put(32768, 16), w.put(32768 ^ 0xffff, 16), w.raw(head.tobytes())
    w.put(1, 1), w.put(1, 2)
    for _ in range(3):
        w.put(0b11000101, 8, True)   # length symbol 285 (258): fixed code 11000101
        w.put(29, 5, True)           # distance symbol 29: 24 577 + 13 extra bits
        w.put(32768 - 24577, 13)
    w.put(0x30 + ord("Z"), 8, True)
    w.put(0, 7, True)
    hand = np.frombuffer(w.bytes(), np.uint8)
    expect = np.concatenate([head, head[: 3 * 258], np.frombuffer(b"Z", np.uint8)])
    assert zlib.decompress(hand.tobytes(), -15) == expect.tobytes()
    check(backend, "Deflate", chunks + [expect], comp + [hand])


def test_large_chunk(backend):
    """The decoder has no 64 KiB limit (only the compressor: benchmarks/benchmark_deflate_chunked.cu:53-63)."""
    n = (1 << 20) if backend.name == "gpu" else 200000
    data = datasets.silesia_style(n, 9)
    check(backend, "Deflate", [data], [raw_deflate(data, 6)])


def test_unchecked_and_without_sizes(backend):
    """device_statuses == NULL and device_actual_uncompressed_bytes == NULL (doc/lowlevel_c_quickstart.md:140)."""
    chunks = datasets.split_chunks(datasets.silesia_style(2 * 65536, 4))
    comp = [raw_deflate(c, 6) for c in chunks]
    check(backend, "Deflate", chunks, comp, checked=False)
    check(backend, "Deflate", chunks, comp, want_actual=False)
    check(backend, "Deflate", chunks, comp, checked=False, want_actual=False)


def test_get_decompress_size(backend):
    """DEFLATE carries no length: the size query decodes and counts (doc/lowlevel_c_quickstart.md:96-109)."""
    from nvcomp_amd.batched import make_batch

    chunks = datasets.split_chunks(datasets.silesia_style(3 * 65536 + 777, 6)) + [np.zeros(0, np.uint8)]
    comp = [raw_deflate(c, l) for c, l in zip(chunks, (9, 1, 0, 6, 6))]
    comp.append(np.frombuffer(b"\x07\xff\xff", np.uint8))  # block type 3: not a stream
    d = backend.dev
    codec = backend.codec("Deflate")
    cb = make_batch(d, comp, align=1)
    sizes = d.upload(np.full(len(comp), 0xDEADBEEF, dtype=np.uint64).view(np.uint8))
    assert codec.get_decompress_size_async(cb, sizes) == 0
    d.synchronize()
    assert d.download(sizes).view(np.uint64)[: len(comp)].tolist() == [c.size for c in chunks] + [0]


def test_output_capacity_is_respected(backend):
    chunks = datasets.split_chunks(datasets.CLASSES["text"](2 * 65536, 3))
    comp = [raw_deflate(c, 6) for c in chunks]
    codec = backend.codec("Deflate")
    outs, actual, status = codec.decompress(comp, [65536, 60000])  # the canary behind each slot is checked inside
    assert status.tolist() == [NvcompStatus.Success, NvcompStatus.ErrorCannotDecompress]
    assert actual.tolist() == [65536, 0]
    assert np.array_equal(outs[0], chunks[0])


def test_malformed_streams_fail_cleanly(backend):
    """Truncations, bit flips and hand-made illegal headers: a chunk either fails (size 0, CannotDecompress) or decodes
    to something -- never writes past its slot (canary), never hangs."""
    rng = np.random.RandomState(5)
    base = datasets.CLASSES["text"](30000, 7)
    good = raw_deflate(base, 6)
    bad = [good[: good.size // 2], good[:3], good[:1]]
    for _ in range(12 if backend.name == "gpu" else 6):
        g = good.copy()
        at = rng.randint(0, g.size, 3)
        g[at] ^= rng.randint(1, 256, 3).astype(np.uint8)
        bad.append(g)
    bad += [np.frombuffer(b, np.uint8) for b in (
        b"\x07",                                    # BTYPE 3
        b"\x01\x05\x00\x00\x00hello",               # stored, LEN / NLEN do not agree
        b"\x01\x05\x00\xfa\xffhel",                 # stored, shorter than LEN
        b"\x05\xff\xff\xff\xff\xff\xff\xff\xff",    # dynamic header with HLIT = 31 (> 286 symbols)
        b"\x03\x00",                                # fixed block: end of block at once -> valid, empty
        b"\x63\x00\x00",                            # fixed block without an end
        b"\x4b\x04\x02",                            # fixed: 'a', then a match 1 back of ... before anything else?
    )]
    codec = backend.codec("Deflate")
    outs, actual, status = codec.decompress(bad, [40000] * len(bad))
    for i, (cc, st, n) in enumerate(zip(bad, status.tolist(), actual.tolist())):
        try:
            ref = zlib.decompress(cc.tobytes(), -15)
        except zlib.error:
            ref = None
        if ref is None:
            # (zlib also rejects streams with trailing garbage or an incomplete code set; we may accept a superset,
            # but what we accept must at least be what a bit-serial reading of RFC 1951 yields)
            assert st in (NvcompStatus.Success, NvcompStatus.ErrorCannotDecompress), (i, st)
            assert n == 0 or st == NvcompStatus.Success
        else:
            assert st == NvcompStatus.Success and n == len(ref), (i, st, n, len(ref))
            assert outs[i][:n].tobytes() == ref
    assert status[0] == NvcompStatus.ErrorCannotDecompress and status[1] == NvcompStatus.ErrorCannotDecompress
    assert status.tolist()[-7:-3] == [NvcompStatus.ErrorCannotDecompress] * 4
    assert status.tolist()[-3] == NvcompStatus.Success and actual.tolist()[-3] == 0


def test_match_before_start_is_an_error(backend):
    """A distance reaching behind the first output byte (zlib: 'invalid distance too far back')."""
    o = zlib.compressobj(9, zlib.DEFLATED, -15, 8, zlib.Z_FIXED)
    good = o.compress(b"abcabcabcabcabcabc") + o.flush()
    assert zlib.decompress(good, -15) == b"abcabcabcabcabcabc"
    # the same stream decoded with nothing before it is fine; cut the three literals off by hand-building:
    # fixed block, literal 'a' (0x61 -> code 0x91, 8 bits), match length 3 distance 4 (> 1 byte produced)
    w = BitWriter()
    w.put(1, 1), w.put(1, 2)            # BFINAL, fixed
    w.put(0x30 + 0x61, 8, True)         # literal 'a'
    w.put(0b0000001, 7, True)           # length symbol 257 (length 3)
    w.put(3, 5, True)                   # distance symbol 3 (distance 4)
    w.put(0, 7, True)                   # end of block
    raw = w.bytes()
    with pytest.raises(zlib.error):
        zlib.decompress(bytes(raw), -15)
    codec = backend.codec("Deflate")
    outs, actual, status = codec.decompress([np.frombuffer(bytes(raw), np.uint8)], [64])
    assert status.tolist() == [NvcompStatus.ErrorCannotDecompress] and actual.tolist() == [0]


def test_gzip_members(backend):
    """Header with and without optional fields, ISIZE checked, size query from the trailer."""
    from nvcomp_amd.batched import make_batch

    data = datasets.silesia_style(3 * 65536, 8)
    chunks = datasets.split_chunks(data) + [np.zeros(0, np.uint8), np.frombuffer(b"x", np.uint8)]
    comp = [gzip_member(c) for c in chunks]
    buf = io.BytesIO()
    with gzip.GzipFile(filename="a_file_name.txt", mode="wb", fileobj=buf, mtime=1234567) as f:  # FNAME set
        f.write(chunks[1].tobytes())
    comp.append(np.frombuffer(buf.getvalue(), np.uint8))
    chunks.append(chunks[1])
    # FEXTRA + FCOMMENT + FHCRC, written by hand around a raw stream
    raw = raw_deflate(chunks[2], 6).tobytes()
    head = b"\x1f\x8b\x08" + bytes([4 | 16 | 2]) + b"\0\0\0\0\0\xff" + b"\x05\x00EXTRA" + b"a comment\0" + b"\x12\x34"
    tail = zlib.crc32(chunks[2].tobytes()).to_bytes(4, "little") + (chunks[2].size & 0xffffffff).to_bytes(4, "little")
    comp.append(np.frombuffer(head + raw + tail, np.uint8))
    chunks.append(chunks[2])
    check(backend, "Gzip", chunks[:-1], comp[:-1])
    codec = backend.codec("Gzip")
    outs, actual, status = codec.decompress(comp, [max(c.size, 1) for c in chunks])
    assert (status == 0).all() and actual.tolist() == [c.size for c in chunks]
    assert np.array_equal(outs[-1], chunks[-1])
    # the size query
    d = backend.dev
    cb = make_batch(d, comp, align=1)
    sizes = d.upload(np.zeros(len(comp), dtype=np.uint64).view(np.uint8))
    assert codec.get_decompress_size_async(cb, sizes) == 0
    d.synchronize()
    assert d.download(sizes).view(np.uint64)[: len(comp)].tolist() == [c.size for c in chunks]
    # a wrong ISIZE, a wrong magic, a raw stream handed to the gzip entry point
    wrong = comp[0].copy()
    wrong[-4] ^= 1
    magic = comp[0].copy()
    magic[1] = 0x8c
    outs, actual, status = codec.decompress([wrong, magic, raw_deflate(chunks[0], 6)], [65536] * 3)
    assert status.tolist() == [NvcompStatus.ErrorCannotDecompress] * 3 and actual.tolist() == [0, 0, 0]


def test_compressor_round_trip(backend):
    """nvcompBatchedDeflateCompressAsync writes standard streams (examples/deflate_cpu_decompression.cu:93-170: the
    CPU inflaters must read them) within the declared bound; our decoder reads them back."""
    chunks = datasets.split_chunks(datasets.silesia_style(2 * 65536 + 100, 2)) + [np.zeros(0, np.uint8), np.frombuffer(b"q", np.uint8)]
    codec = backend.codec("Deflate")
    comp = codec.compress(chunks)
    for cc, c in zip(comp, chunks):
        assert zlib.decompress(cc.tobytes(), -15) == c.tobytes()
    check(backend, "Deflate", chunks, comp)
    assert backend.codec("Deflate").compress([chunks[0]], max_chunk=1000)[0].size == 0  # larger than declared: refused


@pytest.mark.parametrize("algo", [0, 1, 2])
@pytest.mark.parametrize("name", CLASSES)
def test_compressor_classes(backend, name, algo):
    """algo 0: LZ77 + the fixed Huffman code; algo 1, 2: per-chunk (dynamic) codes. Every class decodes with zlib,
    compresses at least as well as the byte-oriented LZ formats do (dynamic codes: about what zlib level 1 reaches),
    and never expands by more than the stored form's five bytes per block."""
    size = 4 * 65536 if backend.name == "gpu" else 65536 + 3333
    data = datasets.CLASSES[name](size, 6)
    chunks = datasets.split_chunks(data)
    codec = backend.codec("Deflate", (algo,))
    comp = codec.compress(chunks)
    for cc, c in zip(comp, chunks):
        assert zlib.decompress(cc.tobytes(), -15) == c.tobytes()
        assert cc.size <= c.size + 5 * (c.size // 65535 + 1)
    ratio = data.size / sum(c.size for c in comp)
    floor = {"text": 1.9, "table": 1.7, "float_csv": 1.5, "float32": 1.5, "int32": 30.0, "lowcard": 1.7, "zeros": 100.0, "noise": 0.999}[name]
    if algo:
        floor = {"text": 2.5, "table": 2.2, "float_csv": 2.2, "float32": 2.0, "int32": 40.0, "lowcard": 2.8, "zeros": 200.0, "noise": 0.999}[name]
    assert ratio >= floor, ratio
    check(backend, "Deflate", chunks, comp)


@pytest.mark.parametrize("algo", [0, 1])
def test_compressor_edges(backend, algo):
    """Chunks shorter than a window, runs far longer than the 258 a match may have (cut into pieces of at least 3),
    literal runs of thousands of bytes in front of a match, matches at the 32 768 limit of the format."""
    rng = np.random.RandomState(23)
    noise = rng.randint(0, 256, 40000, dtype=np.uint8)
    text = datasets.CLASSES["text"](70000, 2)
    chunks = [np.frombuffer(b"abcabcabcabcabcabcabcabc", np.uint8), text[:15], text[:16], text[:17], text[:63], text[:64], text[:65],
              np.full(65536, 0x90, np.uint8), np.full(259 + 3, 7, np.uint8), np.full(258 + 258 + 2, 9, np.uint8),
              np.concatenate([noise[:30000], text[:2000], noise[:30000]]),           # far match: distance 32 000
              np.concatenate([noise[:33000], noise[:32000]]),                         # distance 33 000: out of reach
              np.concatenate([noise[:5000], np.zeros(300, np.uint8), noise[5000:9000]]),
              np.frombuffer(bytes(range(256)) * 8, np.uint8),                          # every literal once per 256: flat counts
              np.concatenate([np.full(60000, 65, np.uint8), rng.randint(0, 256, 5536, dtype=np.uint8)]),  # one symbol dominates
              text[:1023], text[:1024], text[:1025]]                                   # around the dynamic coder's size threshold
    codec = backend.codec("Deflate", (algo,))
    comp = codec.compress(chunks)
    for cc, c in zip(comp, chunks):
        assert zlib.decompress(cc.tobytes(), -15) == c.tobytes()
        assert cc.size <= c.size + 5 * (c.size // 65535 + 1)
    assert comp[7].size < 500 and comp[10].size < 40000  # the run; most of the repeat 32 000 back is found
    check(backend, "Deflate", chunks, comp)


def test_dynamic_codes_on_skewed_alphabets(backend):
    """Code construction under stress: byte distributions from flat to Fibonacci-steep (code lengths that want more than
    15 bits and must be repaired), few and many distinct symbols, with and without matches. zlib must read every block."""
    rng = np.random.RandomState(99)
    chunks = []
    for i in range(40 if backend.name == "gpu" else 10):
        n = int(rng.choice([1500, 5000, 20000, 65536]))
        k = int(rng.choice([2, 3, 17, 64, 200, 256]))
        steep = float(rng.choice([0.0, 0.3, 0.62, 1.0, 2.0]))  # p(symbol j) ~ exp(-steep * j): 0.62 ~ golden ratio decay
        p = np.exp(-steep * np.arange(k))
        c = rng.choice(k, size=n, p=p / p.sum()).astype(np.uint8)
        if i % 3 == 0:
            c[n // 2:] = c[: n - n // 2]  # long-distance repeats on top
        chunks.append(c)
    codec = backend.codec("Deflate", (1,))
    comp = codec.compress(chunks)
    for cc, c in zip(comp, chunks):
        assert zlib.decompress(cc.tobytes(), -15) == c.tobytes()
        assert cc.size <= c.size + 5 * (c.size // 65535 + 1)
    check(backend, "Deflate", chunks, comp)
