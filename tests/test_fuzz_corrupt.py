"""Corrupted-stream fuzz for every decoder: random bit flips, byte splats, truncations and extensions of valid
streams. A decoder may accept or reject a damaged stream, but it must never write outside the capacity it was
given (canary bytes behind every output slot), must agree with the CPU oracle on accept/reject for the
own-format codecs, and, when it accepts, must report a size within the capacity. The reference documents the
same contract: invalid data yields a per-chunk status, never a crash (doc/lowlevel_c_quickstart.md:140,
CHANGELOG.md:160-164)."""
import numpy as np
import pytest

from nvcomp_amd import datasets
from nvcomp_amd._lib import NvcompStatus


def damage(rng, stream):
    b = stream.copy()
    kind = rng.randint(0, 6)
    if b.size == 0:
        return b
    if kind == 0:  # single bit flips
        for _ in range(int(rng.randint(1, 4))):
            b[rng.randint(0, b.size)] ^= 1 << rng.randint(0, 8)
    elif kind == 1:  # splat a run of random bytes
        at = rng.randint(0, b.size)
        n = min(int(rng.choice([1, 2, 4, 16, 64])), b.size - at)
        b[at:at + n] = rng.randint(0, 256, size=n)
    elif kind == 2:  # truncate
        b = b[: rng.randint(0, b.size)]
    elif kind == 3:  # header damage
        b[rng.randint(0, min(16, b.size))] = rng.randint(0, 256)
    elif kind == 4:  # 0xFF run (length-extension bytes of the LZ formats)
        at = rng.randint(0, b.size)
        n = min(int(rng.choice([2, 8, 300])), b.size - at)
        b[at:at + n] = 0xFF
    else:  # garbage appended
        b = np.concatenate([b, rng.randint(0, 256, size=int(rng.choice([1, 7, 100]))).astype(np.uint8)])
    return b


def valid_streams(oracle, fmt, chunks):
    if fmt == "LZ4":
        return [oracle.lz4_compress(c) for c in chunks], oracle.lz4_decompress, None
    if fmt == "Snappy":
        return [oracle.snappy_compress(c) for c in chunks], oracle.snappy_decompress, None
    if fmt == "Cascaded":
        return [oracle.cascaded_compress(c, 4096, 5, 2, 1, 1) for c in chunks], oracle.cascaded_decompress, (4096, 5, 2, 1, 1)
    if fmt == "Bitcomp":
        return [oracle.bitcomp_compress(c, 0, 2) for c in chunks], oracle.bitcomp_decompress, (0, 3)
    return [oracle.ans_compress(c) for c in chunks], oracle.ans_decompress, (0,)


@pytest.mark.parametrize("fmt", ["LZ4", "Snappy", "Cascaded", "Bitcomp", "ANS"])
def test_corrupt_streams_are_contained(backend, lz_path, oracle, fmt):
    rng = np.random.RandomState(["LZ4", "Snappy", "Cascaded", "Bitcomp", "ANS"].index(fmt) * 101 + 17)
    n = 96 if backend.name == "gpu" else 24
    # float_columns: short runs in both Cascaded layers (the decoder's rle_expand_direct / rle_expand_inplace path)
    gens = [datasets.text, datasets.int32_column, datasets.lowcard, datasets.table_rows, datasets.float_columns]
    chunks = [gens[i % 5](int(rng.choice([600, 4096, 20000, 65536])), i) for i in range(n)]
    good, dec, opts = valid_streams(oracle, fmt, chunks)
    bad = [damage(rng, g) for g in good]
    caps = [c.size for c in chunks]
    codec = backend.codec(fmt, opts)
    align = 8 if fmt == "Cascaded" else 1
    outs, actual, status = codec.decompress(bad, caps, comp_align=align, out_align=align)  # canaries checked inside
    own = fmt in ("Cascaded", "Bitcomp", "ANS")
    for i, (b, cap) in enumerate(zip(bad, caps)):
        rc, ref = dec(b, cap)
        if status[i] == NvcompStatus.Success:
            assert actual[i] <= cap
            if own or rc == 0:
                # the own-format models are exact about validity; for LZ4/Snappy an accepted stream must decode alike
                assert rc == 0, f"{fmt} chunk {i}: HIP decoder accepted a stream the CPU oracle rejects"
                assert actual[i] == ref.size and np.array_equal(outs[i][: ref.size], ref)
        else:
            assert actual[i] == 0
            if own:
                assert rc != 0, f"{fmt} chunk {i}: HIP decoder rejected a stream the CPU oracle accepts"
    if own:
        # these decoders validate regardless of `device_statuses`; for LZ4/Snappy a NULL status array turns the
        # bounds checks off by contract ("OOB error checking is disabled", doc/lowlevel_c_quickstart.md:140)
        codec.decompress(bad, caps, checked=False, comp_align=align, out_align=align)


@pytest.mark.parametrize("fmt", ["LZ4", "Snappy"])
def test_corrupt_column_streams_are_contained(backend, lz_path, oracle, fmt):
    """The same on the data the run executor takes (common/lz_window.hip.h: execute_run_batch; chunks that shrank 8 x go
    through the loop that holds it): sorted-key and int32 columns through liblz4 (default and HC) / libsnappy, damaged --
    offsets that stop being multiples of the period, lengths that overshoot the chunk, periods that change under a run,
    a speculated match whose source moved. Canaries behind every slot; an accepted stream decodes like the CPU oracle's."""
    rng = np.random.RandomState(77 if fmt == "LZ4" else 78)
    n = 120 if backend.name == "gpu" else 30
    gens = [datasets.mortgage_col0_like, datasets.int32_column]
    chunks = [gens[i % 2](65536 if i % 3 else int(rng.choice([9000, 30000])), i) for i in range(n)]
    if fmt == "LZ4":
        good = [(oracle.ref_lz4_compress(c, 12 if i % 4 == 0 else 0) if oracle.have_ref() else oracle.lz4_compress(c)) for i, c in enumerate(chunks)]
        dec = oracle.lz4_decompress
    else:
        good = [(oracle.ref_snappy_compress(c) if oracle.have_ref() else oracle.snappy_compress(c)) for c in chunks]
        dec = oracle.snappy_decompress
    bad = [damage(rng, g) if i % 8 else g for i, g in enumerate(good)]  # one in eight stays valid
    caps = [c.size for c in chunks]
    codec = backend.codec(fmt)
    outs, actual, status = codec.decompress(bad, caps)  # canaries checked inside
    accepted = 0
    for i, (b, cap) in enumerate(zip(bad, caps)):
        rc, ref = dec(b, cap)
        if status[i] == NvcompStatus.Success:
            accepted += 1
            assert actual[i] <= cap
            if rc == 0:
                assert actual[i] == ref.size and np.array_equal(outs[i][: ref.size], ref), f"{fmt} chunk {i}"
        else:
            assert actual[i] == 0
            assert rc != 0, f"{fmt} chunk {i}: rejected a stream the CPU oracle reads"
        if i % 8 == 0:
            assert status[i] == NvcompStatus.Success and np.array_equal(outs[i], chunks[i]), f"{fmt} chunk {i}: a valid stream"
    assert 0 < accepted < n


@pytest.mark.parametrize("fmt", ["Deflate", "Gzip"])
def test_corrupt_deflate_streams_are_contained(backend, fmt):
    """The same for DEFLATE / gzip against zlib: what zlib accepts must decode to the same bytes; what it rejects may be
    rejected or (trailing garbage, incomplete code sets zlib is stricter about) accepted -- never a write past the slot,
    never a hang."""
    import zlib

    rng = np.random.RandomState(4242 if fmt == "Deflate" else 2424)
    n = 160 if backend.name == "gpu" else 30
    gens = [datasets.text, datasets.int32_column, datasets.lowcard, datasets.table_rows, datasets.float_columns]
    chunks = [gens[i % 5](int(rng.choice([300, 4096, 20000, 65536])), i) for i in range(n)]
    wbits = -15 if fmt == "Deflate" else 15 | 16
    good = []
    for i, c in enumerate(chunks):
        o = zlib.compressobj([1, 6, 9][i % 3], zlib.DEFLATED, wbits, 8, [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY][(i // 3) % 3])
        good.append(np.frombuffer(o.compress(c.tobytes()) + o.flush(), dtype=np.uint8))
    bad = [damage(rng, g) for g in good]
    caps = [c.size for c in chunks]
    codec = backend.codec(fmt)
    outs, actual, status = codec.decompress(bad, caps)  # canaries checked inside
    accepted = 0
    for i, (b, cap) in enumerate(zip(bad, caps)):
        try:
            d = zlib.decompressobj(wbits)
            ref = d.decompress(b.tobytes(), cap + 1)
            ok = d.eof and len(ref) <= cap
        except zlib.error:
            ok = False
        if status[i] == NvcompStatus.Success:
            accepted += 1
            assert actual[i] <= cap
            if ok:
                assert actual[i] == len(ref) and outs[i][: len(ref)].tobytes() == ref, f"{fmt} chunk {i}"
        else:
            assert actual[i] == 0
            assert not ok or fmt == "Gzip", f"{fmt} chunk {i}: rejected a stream zlib reads"  # (gzip: we also check ISIZE of cut members)
    assert 0 < accepted < n
    # device_statuses == NULL must not turn the bounds / match-offset checks off (these streams come from outside;
    # only the status write is optional): same canaries, same sizes, same bytes where the stream was accepted
    outs2, actual2, status2 = codec.decompress(bad, caps, checked=False)
    assert status2 is None and actual2.tolist() == actual.tolist()
    for i in range(n):
        if status[i] == NvcompStatus.Success:
            assert np.array_equal(outs2[i][: actual[i]], outs[i][: actual[i]])
