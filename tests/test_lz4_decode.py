"""LZ4 batched decompress parity: HIP path (or its host emulation) vs the CPU oracle.

Mirrors the reference's own format pin examples/lz4_cpu_compression.cu:59-74,137
(CPU-compressed chunks -> nvcompBatchedLZ4DecompressAsync -> byte compare) and the
round-trip checks of benchmarks/benchmark_template_chunked.cuh:553-584
(status == nvcompSuccess, actual size == original size, bytes identical).
"""
import numpy as np
import pytest

from nvcomp_amd import datasets
from nvcomp_amd._lib import NvcompStatus


def cpu_compress(oracle, chunks, hc=0):
    if oracle.have_ref():
        return [oracle.ref_lz4_compress(c, hc) for c in chunks]
    return [oracle.lz4_compress(c) for c in chunks]


def check_roundtrip(backend, oracle, chunks, comp, **kw):
    codec = backend.codec("LZ4")
    caps = [c.size for c in chunks]
    outs, actual, status = codec.decompress(comp, caps, **kw)
    if status is not None:
        assert (status == NvcompStatus.Success).all(), status
    if actual is not None:
        assert actual.tolist() == caps
    for i, (o, c) in enumerate(zip(outs, chunks)):
        assert np.array_equal(o, c), f"chunk {i} differs"
    # the oracle agrees on every chunk
    for cc, c in zip(comp, chunks):
        rc, ref = oracle.lz4_decompress(cc, c.size)
        assert rc == 0 and np.array_equal(ref, c)


@pytest.mark.parametrize("name", ["text", "table", "float_csv", "float32", "int32", "lowcard", "zeros", "noise"])
def test_classes_small(backend, lz_path, oracle, name):
    size = 3 * 65536 + 1234 if backend.name == "gpu" else 65536 + 777
    data = datasets.CLASSES[name](size, 1)
    chunks = datasets.split_chunks(data)
    check_roundtrip(backend, oracle, chunks, cpu_compress(oracle, chunks))


def test_hc_compressed_input(backend, lz_path, oracle):
    if not oracle.have_ref():
        pytest.skip("liblz4 not available")
    data = datasets.table_rows(2 * 65536, 3)
    chunks = datasets.split_chunks(data)
    check_roundtrip(backend, oracle, chunks, cpu_compress(oracle, chunks, hc=12))


@pytest.mark.parametrize("checked,want_actual", [(True, True), (False, True), (True, False), (False, False)])
def test_nullable_outputs(backend, lz_path, oracle, checked, want_actual):
    data = datasets.text(40000, 5)
    chunks = datasets.split_chunks(data, 16384)
    check_roundtrip(backend, oracle, chunks, cpu_compress(oracle, chunks), checked=checked, want_actual=want_actual)


def test_ragged_and_tiny_chunks(backend, lz_path, oracle):
    rng = np.random.RandomState(7)
    sizes = [0, 1, 2, 3, 4, 5, 11, 12, 13, 14, 15, 16, 17, 31, 63, 64, 65, 255, 256, 257, 1000, 4095, 4097]
    base = datasets.text(8192, 9)
    chunks = [base[rng.randint(0, 2000):][:s].copy() for s in sizes]
    comp = cpu_compress(oracle, chunks)
    # liblz4 turns an empty input into a 1-byte block; also feed a true 0-byte chunk
    check_roundtrip(backend, oracle, chunks, comp)
    chunks2 = [np.zeros(0, np.uint8), base[:100].copy()]
    comp2 = [np.zeros(0, np.uint8), comp[0][:0]]
    comp2[1] = cpu_compress(oracle, [chunks2[1]])[0]
    check_roundtrip(backend, oracle, chunks2, comp2)


def test_unaligned_everything(backend, lz_path, oracle):
    data = datasets.table_rows(30000, 11)
    chunks = datasets.split_chunks(data, 5000)
    comp = cpu_compress(oracle, chunks)
    for mis in (1, 2, 3):
        check_roundtrip(backend, oracle, chunks, comp, base_misalign=mis)


def test_long_matches_and_overlaps(backend, lz_path, oracle):
    # periods 1..70 and a few long ones exercise the pattern-doubling match copy
    parts = []
    rng = np.random.RandomState(13)
    for period in list(range(1, 71)) + [100, 255, 256, 257, 1000, 1024, 1025, 3000]:
        pat = rng.randint(0, 256, size=period).astype(np.uint8)
        reps = max(2, (rng.randint(50, 3000) // period) + 2)
        parts.append(np.tile(pat, reps))
        parts.append(rng.randint(0, 256, size=rng.randint(1, 40)).astype(np.uint8))
    data = np.concatenate(parts)
    chunks = datasets.split_chunks(data)
    check_roundtrip(backend, oracle, chunks, cpu_compress(oracle, chunks))


def test_long_overlapping_matches_inside_text(backend, lz_path, oracle):
    """Runs and short periods between stretches of text: the chunk's ratio stays far below the 16 x from which the
    workgroup-per-chunk decoder hands a chunk to its two-wave fallback (common/lz_team.hip.h), so that its OWN handling of
    matches that overlap their source (periods below the match length, the step's frontier) is what runs here."""
    rng = np.random.RandomState(21)
    noise = rng.randint(0, 256, size=1 << 16).astype(np.uint8)
    parts = []
    for period in [1, 2, 3, 4, 5, 7, 8, 12, 16, 31, 64, 100, 255]:
        for _ in range(3):
            pat = rng.randint(0, 256, size=period).astype(np.uint8)
            parts.append(np.tile(pat, rng.randint(40, 900) // period + 2))
            a = rng.randint(0, 60000)
            parts.append(noise[a: a + rng.randint(150, 400)])
    data = np.concatenate(parts)
    chunks = datasets.split_chunks(data)
    comp = cpu_compress(oracle, chunks)
    assert all(16 * cc.size > c.size for cc, c in zip(comp, chunks)), "the chunks must stay team chunks"
    check_roundtrip(backend, oracle, chunks, comp)


def _lz4_block(seqs, tail):
    """Hand-built LZ4 block: seqs = [(literal bytes, offset, match length >= 4)], tail = the final literal-only sequence."""
    out = bytearray()

    def lens(n):
        b = bytearray()
        while n >= 255:
            b.append(255)
            n -= 255
        b.append(n)
        return b

    for lit, off, mlen in seqs:
        ll, ml = len(lit), mlen - 4
        out.append((min(ll, 15) << 4) | min(ml, 15))
        if ll >= 15:
            out += lens(ll - 15)
        out += lit
        out += bytes([off & 255, off >> 8])
        if ml >= 15:
            out += lens(ml - 15)
    ll = len(tail)
    out.append(min(ll, 15) << 4)
    if ll >= 15:
        out += lens(ll - 15)
    out += tail
    return np.frombuffer(bytes(out), dtype=np.uint8)


def _lz4_expand(seqs, tail):
    out = bytearray()
    for lit, off, mlen in seqs:
        out += lit
        for _ in range(mlen):
            out.append(out[-off])
    out += tail
    return np.frombuffer(bytes(out), dtype=np.uint8)


def test_streamed_long_runs(backend, lz_path, oracle):
    """Long literal runs and long matches leave the window path and are written straight to the output buffer
    (common/lz_window.hip.h: stream_sequence): every period class (a register tile that never changes, one that rotates,
    HBM-to-HBM copies whose effective offset doubles), lengths around the thresholds and the step sizes, every output
    alignment, short sequences in between (the window restarts behind a streamed one)."""
    rng = np.random.RandomState(77)
    offs = [1, 2, 3, 4, 5, 7, 8, 13, 16, 31, 32, 64, 100, 255, 256, 257, 300, 1000, 1023, 1024, 1025, 1500, 4095, 4096, 4097, 5200, 9000]
    lens = [127, 128, 129, 200, 255, 256, 257, 300, 1023, 1024, 1025, 1040, 4095, 4096, 4113, 5000, 20000]
    if backend.name == "emu":
        offs, lens = offs[::2] + [256, 1024], lens[::2] + [128, 4096]
    blocks, raws = [], []
    for i, off in enumerate(offs):
        seqs = [(rng.randint(0, 256, size=max(off, 8) + i % 5).astype(np.uint8).tobytes(), min(off, 4), 4)]  # something to point at
        seqs[0] = (seqs[0][0], 1 + (i % 3), 4 + i % 7)
        for j, mlen in enumerate(lens):
            lit = rng.randint(0, 256, size=(j * 7 + i) % 23).astype(np.uint8).tobytes()
            produced = sum(len(l) + m for l, _, m in seqs)
            if off > produced + len(lit):
                lit += rng.randint(0, 256, size=off - produced - len(lit)).astype(np.uint8).tobytes()
            seqs.append((lit, off, mlen))
            seqs.append((rng.randint(0, 256, size=j % 4).astype(np.uint8).tobytes(), 1 + j % 9, 4 + j % 13))  # a short one behind it
        tail = rng.randint(0, 256, size=5 + i % 11).astype(np.uint8).tobytes()
        blocks.append(_lz4_block(seqs, tail))
        raws.append(_lz4_expand(seqs, tail))
    # long literal runs: lengths around the threshold and the 1 KiB / 4 KiB steps, matches in between
    for n in [255, 256, 257, 1000, 1024, 1039, 4096, 4111, 5000, 40000]:
        lit = rng.randint(0, 256, size=n).astype(np.uint8).tobytes()
        seqs = [(b"abc", 1, 9), (lit, 7, 30), (lit[: n // 3], n // 2 + 1, 500), (b"", 3, 4)]
        blocks.append(_lz4_block(seqs, lit[:300]))
        raws.append(_lz4_expand(seqs, lit[:300]))
    for cc, c in zip(blocks, raws):
        rc, ref = oracle.lz4_decompress(cc, c.size)
        assert rc == 0 and np.array_equal(ref, c), "the hand-built block is not what the oracle reads"
    for mis in (0, 1, 5, 15):
        check_roundtrip(backend, oracle, raws, blocks, base_misalign=mis)


def _run_blocks(rng, n_seqs, offs, lit_choices, mlen_choices, first_lit=16):
    """A chain of run sequences: (a few literals, a match of period `off`), the shapes lzw::execute_run_batch takes."""
    seqs = [(rng.randint(0, 256, size=first_lit).astype(np.uint8).tobytes(), offs[0], 20)]
    for j in range(n_seqs):
        off = offs[j % len(offs)]
        nlit = lit_choices[rng.randint(len(lit_choices))]
        mlen = mlen_choices[rng.randint(len(mlen_choices))]
        seqs.append((rng.randint(0, 256, size=nlit).astype(np.uint8).tobytes(), off, mlen))
    return seqs


def test_run_batches(backend, lz_path, oracle):
    """Sequences of "a few literals, then a match of period 1, 2, 4, 8 or 16" -- sorted key columns, typed columns -- are
    executed up to 64 at a time straight to the output buffer (common/lz_window.hip.h: execute_run_batch): every period,
    0 .. 16 literals (17 and more end the batch), runs from 16 bytes on (shorter ones end it), sequences without literals
    that continue the run in front of them (merged), a period that changes, batches at every alignment of the output
    buffer, at the start and at the end of a chunk, and behind / in front of ordinary sequences."""
    rng = np.random.RandomState(606)
    blocks, raws = [], []
    few = backend.name == "emu"
    lits_all = list(range(0, 17))
    for off in (1, 2, 4, 8, 16):
        # the column shape: one period, short literals, long runs
        for lit_choices, mlen_choices in (([2], [142, 254, 398, 542, 654]), ([1], [15, 16, 17, 59, 139, 275, 583]),
                                          (lits_all, [16, 17, 30, 31, 32, 33, 47, 48, 100, 1000, 1549, 4000]),
                                          ([0, 0, 1, 2, 3], [4, 5, 6, 7, 12, 15, 16, 200, 392, 616]),
                                          ([0, 4, 8, 15, 16, 17, 20], [16, 64, 300])):
            seqs = _run_blocks(rng, 40 if few else 150, [off], lit_choices, mlen_choices, first_lit=max(off, 3) + rng.randint(14))
            tail = rng.randint(0, 256, size=5 + rng.randint(9)).astype(np.uint8).tobytes()
            blocks.append(_lz4_block(seqs, tail))
            raws.append(_lz4_expand(seqs, tail))
    # the period changes inside a chain, ordinary sequences (other offsets, short matches) in between
    for k in range(3 if few else 10):
        seqs = [(rng.randint(0, 256, size=40).astype(np.uint8).tobytes(), 5, 9)]
        for j in range(100 if few else 300):
            kind = rng.randint(10)
            if kind < 7:
                off = (1, 2, 4, 8, 16)[(j // 13 + k) % 5]
                seqs.append((rng.randint(0, 256, size=rng.randint(0, 5)).astype(np.uint8).tobytes(), off, 16 + rng.randint(500)))
            elif kind < 9:
                seqs.append((rng.randint(0, 256, size=rng.randint(0, 20)).astype(np.uint8).tobytes(), 1 + rng.randint(40), 4 + rng.randint(40)))
            else:
                seqs.append((b"", (1, 2, 4, 8, 16)[(j // 13 + k) % 5], 4 + rng.randint(12)))
        tail = rng.randint(0, 256, size=5 + k).astype(np.uint8).tobytes()
        blocks.append(_lz4_block(seqs, tail))
        raws.append(_lz4_expand(seqs, tail))
    # a chunk that starts with runs (the first match needs its period among the first literals), and one that is only runs
    for off in (1, 8, 16):
        seqs = [(rng.randint(0, 256, size=off).astype(np.uint8).tobytes(), off, 300)]
        seqs += [(rng.randint(0, 256, size=2).astype(np.uint8).tobytes(), off, 100 + 7 * j) for j in range(70)]
        blocks.append(_lz4_block(seqs, b"12345"))
        raws.append(_lz4_expand(seqs, b"12345"))
    # the real thing: the sorted key column and the int32 column through liblz4 (default and HC)
    for name in ("mortgage_col0_like", "int32"):
        gen = getattr(datasets, name) if hasattr(datasets, name) else datasets.CLASSES[name]
        data = gen(65536 + 4096, 3)
        for c in datasets.split_chunks(data):
            for hc in ((0, 12) if oracle.have_ref() else (0,)):
                blocks.append(cpu_compress(oracle, [c], hc=hc)[0])
                raws.append(c)
    for cc, c in zip(blocks, raws):
        rc, ref = oracle.lz4_decompress(cc, c.size)
        assert rc == 0 and np.array_equal(ref, c), "the hand-built block is not what the oracle reads"
    for mis in ((0, 7) if few else range(16)):
        check_roundtrip(backend, oracle, raws, blocks, base_misalign=mis)


def test_run_batches_speculated_matches(backend, lz_path, oracle):
    """liblz4's fast compressor starts a run of equal keys with a short match from far back -- the new key's unchanged bytes,
    copied from an earlier key, at a distance that is a multiple of the period. The run executor takes such a match for
    bytes the run would have produced anyway and checks that against the patterns (common/lz_window.hip.h:
    execute_run_batch): matches that hold and matches that do not (the key's high bytes changed in between), sources in
    this batch, in front of it, across a run boundary and inside literals, short and long ones, periods 4, 8 and 16; the
    real column through liblz4; and a distance beyond the output, which must come back as an error, not as bytes."""
    rng = np.random.RandomState(1234)
    few = backend.name == "emu"
    blocks, raws = [], []
    for off in (4, 8, 16):
        for variant in range(3 if few else 8):
            seqs = [(rng.randint(0, 256, size=16).astype(np.uint8).tobytes(), off, 100 + 8 * variant)]
            produced = 116 + 8 * variant
            for j in range(60 if few else 220):
                kind = rng.randint(10)
                far = off * int(rng.randint(2, 1 + max(2, min(produced // off, 120))))
                if kind < 4:  # the fast compressor's pair: literals + a short match from a key further back, then the run
                    nlit = int(rng.randint(1, 4))
                    mlen = int(rng.randint(4, max(5, off - nlit + 1))) if off > 4 else 4
                    seqs.append((rng.randint(0, 256, size=nlit).astype(np.uint8).tobytes(), far, mlen))
                    seqs.append((b"", off, 20 + int(rng.randint(600))))
                elif kind < 6:  # a clean run
                    seqs.append((rng.randint(0, 256, size=int(rng.randint(0, 4))).astype(np.uint8).tobytes(), off, 16 + int(rng.randint(500))))
                elif kind < 7:  # a whole new key: the bytes further back no longer fit
                    seqs.append((rng.randint(0, 256, size=off).astype(np.uint8).tobytes(), off, 16 + int(rng.randint(300))))
                elif kind < 9:  # a long match from far back (a multiple of the period), with and without literals
                    seqs.append((rng.randint(0, 256, size=int(rng.randint(0, 3))).astype(np.uint8).tobytes(), far, 4 + int(rng.randint(60))))
                    seqs.append((b"", off, 16 + int(rng.randint(100))))
                else:  # a short match from far back in front of more literals
                    seqs.append((b"", far, 4 + int(rng.randint(8))))
                    seqs.append((rng.randint(0, 256, size=2).astype(np.uint8).tobytes(), off, 40 + int(rng.randint(200))))
                produced = sum(len(l) + m for l, _, m in seqs)
            tail = rng.randint(0, 256, size=5 + variant).astype(np.uint8).tobytes()
            blocks.append(_lz4_block(seqs, tail))
            raws.append(_lz4_expand(seqs, tail))
    data = datasets.mortgage_col0_like(2 * 65536 + 1000, 5)
    for c in datasets.split_chunks(data):
        blocks.append(cpu_compress(oracle, [c], hc=0)[0])
        raws.append(c)
    for cc, c in zip(blocks, raws):
        rc, ref = oracle.lz4_decompress(cc, c.size)
        assert rc == 0 and np.array_equal(ref, c), "the hand-built block is not what the oracle reads"
    for mis in ((0, 9) if few else range(16)):
        check_roundtrip(backend, oracle, raws, blocks, base_misalign=mis)
    # a distance beyond what has been produced, among runs: an error, whatever path looked at it first
    bad = []
    for where in (0, 1, 5, 30):
        seqs = [(rng.randint(0, 256, size=16).astype(np.uint8).tobytes(), 8, 100)]
        for j in range(40):
            far = 8 * 4000 if j == where else 8
            seqs.append((rng.randint(0, 256, size=2).astype(np.uint8).tobytes(), far, 6 if far != 8 else 200))
            seqs.append((b"", 8, 300))
        bad.append(_lz4_block(seqs, b"12345"))
    codec = backend.codec("LZ4")
    outs, actual, status = codec.decompress(bad, [65536] * len(bad))
    for i, b in enumerate(bad):
        rc, _ = oracle.lz4_decompress(b, 65536)
        assert rc != 0
        assert status[i] != NvcompStatus.Success and actual[i] == 0


def test_two_byte_length_fields(backend, lz_path, oracle):
    """Lengths that take a second, third, ... extension byte -- matches of 274 bytes and more, literal runs of 270 and more
    -- are what a sorted key column compressed by liblz4 consists of (two literals, 170 .. 680 bytes at offset 8, 165 times
    per chunk). The token chase and the batch parser resolve up to six extension bytes of a match length themselves
    (lz4_decode_window.hip.h: DeltaFn::second, parse_batch) and hand everything longer to the general parser: every length
    around every boundary, in long chains (many tokens per 256-byte stream window), mixed with short sequences, and at
    the end of the chunk."""
    rng = np.random.RandomState(4242)
    blocks, raws = [], []
    boundary = [268, 269, 270, 271, 272, 273, 274, 275, 276, 277, 300, 398, 400, 500, 524, 525, 526, 527, 528, 529, 530, 531, 532, 600,
                783, 784, 785, 1038, 1039, 1040, 1293, 1294, 1547, 1548, 1549, 1550, 1803, 1804, 2500]
    # chains of one length class each: the shape of the column
    for k, mlen in enumerate(boundary if backend.name != "emu" else boundary[::3] + [274, 528, 529, 1548, 1549]):
        seqs = [(rng.randint(0, 256, size=16).astype(np.uint8).tobytes(), 8, 40)]
        for j in range(60 if backend.name != "emu" else 14):
            lit = rng.randint(0, 256, size=(j + k) % 4).astype(np.uint8).tobytes()
            seqs.append((lit, (1, 2, 4, 8, 16, 3, 24)[(j + k) % 7], mlen + (j % 3 == 2)))
        tail = rng.randint(0, 256, size=5 + k % 7).astype(np.uint8).tobytes()
        blocks.append(_lz4_block(seqs, tail))
        raws.append(_lz4_expand(seqs, tail))
    # every class mixed, literal runs with two length bytes among them, short sequences in between
    for k in range(6 if backend.name != "emu" else 2):
        seqs = [(rng.randint(0, 256, size=40).astype(np.uint8).tobytes(), 5, 9)]
        for j in range(50 if backend.name != "emu" else 16):
            mlen = boundary[rng.randint(len(boundary))] if j % 3 else 4 + rng.randint(30)
            nlit = (0, 1, 2, 14, 15, 16, 269, 270, 271, 300, 523, 524, 525, 526)[rng.randint(14)] if j % 5 == 4 else rng.randint(6)
            seqs.append((rng.randint(0, 256, size=nlit).astype(np.uint8).tobytes(), 1 + rng.randint(40), mlen))
        tail = rng.randint(0, 256, size=5 + (270 if k % 2 else 0)).astype(np.uint8).tobytes()
        blocks.append(_lz4_block(seqs, tail))
        raws.append(_lz4_expand(seqs, tail))
    for cc, c in zip(blocks, raws):
        rc, ref = oracle.lz4_decompress(cc, c.size)
        assert rc == 0 and np.array_equal(ref, c), "the hand-built block is not what the oracle reads"
    for mis in (0, 3):
        check_roundtrip(backend, oracle, raws, blocks, base_misalign=mis)


def test_stream_ends_on_ring_block_boundaries(backend, lz_path, oracle):
    """The end of a chunk against the stream ring's geometry (common/lz_window.hip.h): the ring is refilled in 1 KiB blocks of
    the chunk's 16-byte-aligned coordinates, bytes behind the chunk's end read as zero inside the last loaded block and as
    OLDER stream bytes when the chunk ends exactly on a block boundary, and the token chase's last windows must not let
    either leak into a token position. Blocks whose last byte lies on, right before and right behind such a boundary, for
    pointers at every offset modulo 16 that matters, ending in short and in long literal runs and in chains of short
    sequences; tiny blocks (a stream shorter than one chase window). (Written for an experiment that let the last windows
    take the straight-line distance function -- neutral on the card, not kept; the cases stay.)"""
    rng = np.random.RandomState(515)

    def block_of(total, style):
        seqs = [(rng.randint(0, 256, size=24).astype(np.uint8).tobytes(), 3, 11)]
        while True:
            body = _lz4_block(seqs, b"12345").size - 6  # without the final literal-only sequence
            room = total - body
            if room < 40:
                break
            produced = sum(len(l) + m for l, _, m in seqs)
            if style == "short":
                lit = rng.randint(0, 256, size=rng.randint(0, 4)).astype(np.uint8).tobytes()
                seqs.append((lit, 1 + rng.randint(min(20, produced + len(lit))), 4 + rng.randint(12)))
            else:
                lit = rng.randint(0, 256, size=rng.randint(5, max(6, min(60, room - 14)))).astype(np.uint8).tobytes()
                seqs.append((lit, 1 + rng.randint(min(200, produced + len(lit))), 4 + rng.randint(300)))
        for _ in range(4):
            for n in range(5, 400):  # the tail literal run that makes the block exactly `total` bytes long
                tail = rng.randint(0, 256, size=n).astype(np.uint8).tobytes()
                b = _lz4_block(seqs, tail)
                if b.size == total:
                    return b, _lz4_expand(seqs, tail)
            # a tail of 15 bytes takes a length byte: one size in 256 cannot be reached -- one more literal in front of it can
            lit, off, mlen = seqs[-1]
            seqs[-1] = (lit + b"x", off, mlen)
        raise AssertionError("no tail length fits")

    # (the emulated workgroup-per-chunk path has no ring and costs 16 x the coroutines: two of the four misalignments)
    for mis in ((0, 15) if backend.name == "emu" and lz_path in ("default", "team") else (0, 1, 7, 15)):
        blocks, raws = [], []
        for k in (1, 2, 3, 5):
            for delta in (-1, 0, 1):
                for style in ("short", "long"):
                    b, r = block_of(1024 * k - mis + delta, style)
                    blocks.append(b)
                    raws.append(r)
        for total in (14, 20, 37, 100, 255, 256, 257):
            b, r = block_of(total + 40, "short")
            blocks.append(b)
            raws.append(r)
        for cc, c in zip(blocks, raws):
            rc, ref = oracle.lz4_decompress(cc, c.size)
            assert rc == 0 and np.array_equal(ref, c), "the hand-built block is not what the oracle reads"
        check_roundtrip(backend, oracle, raws, blocks, comp_align=16, base_misalign=mis)


def test_corrupt_streams_do_not_escape(backend, lz_path, oracle):
    """Invalid input -> status != success and size 0 (CHANGELOG.md:160-164); never a write
    outside the output slot (canaries) and, where the oracle accepts, identical bytes."""
    rng = np.random.RandomState(17)
    data = datasets.text(20000, 21)
    chunks = datasets.split_chunks(data, 4096)
    comp = cpu_compress(oracle, chunks)
    bad, caps = [], []
    for c, raw in zip(comp, chunks):
        b = c.copy()
        kind = rng.randint(0, 4)
        if kind == 0:
            b = b[: rng.randint(1, b.size)]                      # truncated
        elif kind == 1:
            b[rng.randint(0, b.size)] ^= 1 << rng.randint(0, 8)  # bit flip
        elif kind == 2:
            b = np.concatenate([b, rng.randint(0, 256, size=5).astype(np.uint8)])  # trailing junk
        else:
            pass                                                 # valid, but capacity too small
        bad.append(b)
        caps.append(raw.size if kind != 3 else raw.size - 1)
    codec = backend.codec("LZ4")
    outs, actual, status = codec.decompress(bad, caps)
    for i, (b, cap) in enumerate(zip(bad, caps)):
        rc, ref = oracle.lz4_decompress(b, cap)
        if rc == 0:
            assert status[i] == NvcompStatus.Success and actual[i] == ref.size
            assert np.array_equal(outs[i][: ref.size], ref)
        else:
            assert status[i] != NvcompStatus.Success and actual[i] == 0


def test_batch_of_many_small_chunks(backend, oracle):
    """8 200 chunks in one launch: above the two-waves-per-chunk threshold, i.e. the persistent-wave launch as shipped
    (common/lz_launch.hip.h): every chunk must be decoded exactly once."""
    if backend.name != "gpu":
        pytest.skip("8 200 workgroup launches take the emulator half a minute and exercise nothing the smaller batches do not")
    data = datasets.silesia_style(8200 * 384, 3)
    chunks = datasets.split_chunks(data, 384)
    assert len(chunks) >= 8192
    check_roundtrip(backend, oracle, chunks, cpu_compress(oracle, chunks))


@pytest.mark.parametrize("fmt", ["LZ4", "Snappy"])
def test_persistent_workgroups(backend, oracle, fmt):
    """300 small chunks: the eight-wave workgroup-per-chunk launch (257 ... 512 chunks), persistent -- more chunks than
    workgroups stay resident (two on the emulator's one-CU card), the rest drawn from the ticket counter in the temp
    buffer: every chunk decoded exactly once."""
    data = datasets.silesia_style(300 * 160, 5)
    chunks = datasets.split_chunks(data, 160)
    assert 256 < len(chunks) <= 512
    if fmt == "LZ4":
        check_roundtrip(backend, oracle, chunks, cpu_compress(oracle, chunks))
    else:
        comp = [oracle.ref_snappy_compress(c) if oracle.have_ref() else oracle.snappy_compress(c) for c in chunks]
        outs, actual, status = backend.codec("Snappy").decompress(comp, [c.size for c in chunks])
        assert (status == NvcompStatus.Success).all() and actual.tolist() == [c.size for c in chunks]
        assert all(np.array_equal(o, c) for o, c in zip(outs, chunks))


def test_get_decompress_size(backend, oracle):
    data = datasets.silesia_style(4 * 65536, 2, chunk=16384)
    chunks = datasets.split_chunks(data, 16384) + [np.zeros(0, np.uint8)]
    comp = cpu_compress(oracle, chunks)
    sizes = backend.codec("LZ4").get_decompress_size(comp)
    assert sizes.tolist() == [c.size for c in chunks]
    assert sizes.tolist() == [oracle.lz4_decompressed_size(c) for c in comp]


def test_runs_executed_by_the_whole_wave(backend, lz_path, oracle):
    """A few literals and a long match of period 1 .. 16 -- the shape of sorted key columns: trains of such sequences, runs at
    the very start of a chunk, runs whose literals are 0 .. 32 bytes, ordinary sequences in between; a run that ends behind
    the capacity or points in front of the chunk fails the chunk and nothing is written behind the slot."""
    rng = np.random.RandomState(31)
    blocks, raws = [], []
    for off in (1, 2, 4, 8, 16):
        seqs = [(rng.randint(0, 256, size=off).astype(np.uint8).tobytes(), off, 200)]  # the chunk opens with a run
        for j in range(60):
            lit = rng.randint(0, 256, size=(j * 5) % 33).astype(np.uint8).tobytes()
            seqs.append((lit, off, 128 + (j * 37) % 900))
            if j % 7 == 3:
                seqs.append((b"xy", 3, 9))  # an ordinary sequence between two runs
        tail = b"the end"
        blocks.append(_lz4_block(seqs, tail))
        raws.append(_lz4_expand(seqs, tail))
    for cc, c in zip(blocks, raws):
        rc, ref = oracle.lz4_decompress(cc, c.size)
        assert rc == 0 and np.array_equal(ref, c)
    for mis in (0, 3, 9):
        check_roundtrip(backend, oracle, raws, blocks, base_misalign=mis)
    codec = backend.codec("LZ4")
    # capacity 50 bytes short of a run's end; an offset of 16 with 8 bytes produced
    short = _lz4_block([(b"abcdefgh", 8, 400), (b"ij", 8, 300)], b"tail!")
    wrong = _lz4_block([(b"abcdefgh", 16, 400)], b"tail!")
    good = _lz4_block([(b"abcdefgh", 8, 400)], b"tail!")
    outs, actual, status = codec.decompress([short, wrong, good], [8 + 400 + 2 + 250, 1000, 8 + 400 + 5])
    assert status[0] != 0 and status[1] != 0 and status[2] == 0
    assert np.array_equal(outs[2], _lz4_expand([(b"abcdefgh", 8, 400)], b"tail!"))
