"""The token index of the LZ decoders (csrc/common/lz_index.hip.h): 64 joined serial walks must give EXACTLY the chunk's
sequence positions -- a prefix of them, and the offset of the first one left to the classic chase. Checked against a
plain Python walk of the same streams (liblz4 / libsnappy output, hand-built blocks), on the emulator and on the GPU."""
import numpy as np
import pytest

from nvcomp_amd import datasets
from nvcomp_amd.batched import make_batch

LIST_CAP = 64 * 344


def lz4_tokens(s):
    """Token offsets of an LZ4 block (the format: token, literal length bytes, literals, offset, match length bytes)."""
    s = bytes(s)
    n, p, out = len(s), 0, []
    while p < n:
        out.append(p)
        t = s[p]
        q = p + 1
        lit = t >> 4
        if lit == 15:
            while True:
                e = s[q]
                q += 1
                lit += e
                if e != 255:
                    break
        q += lit
        if q >= n:
            break
        q += 2
        if (t & 15) == 15:
            while s[q] == 255:
                q += 1
            q += 1
        p = q
    return out


def run_index(backend, fmt, comp):
    d, lib = backend.dev, backend.lib
    n = len(comp)
    batch = make_batch(d, comp, align=1)
    lists = d.upload(np.zeros(n * LIST_CAP, dtype=np.uint16).view(np.uint8))
    info = d.upload(np.zeros(2 * n, dtype=np.uint32).view(np.uint8))
    fn = getattr(lib, f"nvcompAmdBatched{fmt}TokenIndexAsync")
    assert fn(d.ptr(batch.ptrs), d.ptr(batch.sizes), n, d.ptr(lists), d.ptr(info), d.stream()) == 0
    d.synchronize()
    return d.download(lists).view(np.uint16).reshape(n, LIST_CAP), d.download(info).view(np.uint32).reshape(n, 2)


def check(lists, info, comp, tokens_of, min_cover=None):
    covered = total = 0
    for i, c in enumerate(comp):
        want = tokens_of(c)
        count, resume = int(info[i, 0]), int(info[i, 1])
        got = lists[i, :count].astype(np.int64).tolist()
        # a prefix of the true chain, then the position where the chase takes over: itself on the chain
        assert got == want[:count], f"chunk {i}: the index is not a prefix of the chunk's tokens"
        if count < len(want):
            assert resume == want[count], f"chunk {i}: the chase would resume off the chain ({resume})"
        else:
            assert resume >= c.size or resume == 0 or count == len(want)
        if c.size < 2048 or c.size > 65535:
            assert count == 0 and resume == 0
        covered += count
        total += len(want)
    if min_cover is not None:
        assert covered >= min_cover * total, f"the index covers {covered} of {total} tokens"
    return covered, total


@pytest.mark.parametrize("name", ["silesia_style", "text", "table_rows", "float_csv", "noise", "zeros", "int32_column",
                                  "mortgage_col0_like"])
def test_lz4_index_is_the_token_chain(backend, oracle, name):
    gen = getattr(datasets, name)
    data = gen(6 * 65536, 3)
    chunks = datasets.split_chunks(data)
    for level in (0, 12):
        comp = [oracle.ref_lz4_compress(c, level) for c in chunks]
        lists, info = run_index(backend, "LZ4", comp)
        # text-like data: nearly every token is indexed (all but the chunk's last bytes)
        check(lists, info, comp, lz4_tokens, min_cover=0.97 if name in ("text", "table_rows", "float_csv") else None)


def test_lz4_index_edge_streams(backend, oracle):
    rng = np.random.RandomState(5)
    base = datasets.text(70000, 1)
    comp = []
    # streams around the limits of the index: 2 047 / 2 048 bytes, 65 535 / 65 536, tiny, empty
    for raw_len in (100, 3000, 4500, 4600, 9000, 20000, 65536):
        comp.append(oracle.ref_lz4_compress(base[:raw_len], 0))
    noise = rng.randint(0, 256, 65536).astype(np.uint8)
    comp.append(oracle.ref_lz4_compress(noise, 0))                      # 65 795 bytes: one literal run, too long for the index
    comp.append(oracle.ref_lz4_compress(noise[:65000], 0))              # one literal run, short enough: nothing but a resume at 0
    comp.append(oracle.ref_lz4_compress(np.concatenate([base[:30000], noise[:8000], base[:20000]]), 0))  # a long run in the middle
    comp.append(np.zeros(0, dtype=np.uint8))
    lists, info = run_index(backend, "LZ4", comp)
    check(lists, info, comp, lz4_tokens)
    # the long literal run in the middle ends the index there; everything in front of it is covered
    want = lz4_tokens(comp[-2])
    assert 0 < info[-2, 0] < len(want)


def test_lz4_index_of_garbage_never_leaves_the_buffer(backend):
    """Random bytes as a 'stream': whatever the walks do, the list is increasing, inside the stream, and the call returns."""
    rng = np.random.RandomState(11)
    comp = [rng.randint(0, 256, n).astype(np.uint8) for n in (2048, 5000, 40000, 65535)]
    comp.append(np.full(30000, 0xff, dtype=np.uint8))
    comp.append(np.full(30000, 0x00, dtype=np.uint8))
    comp.append(np.full(30000, 0xf0, dtype=np.uint8))
    lists, info = run_index(backend, "LZ4", comp)
    for i, c in enumerate(comp):
        count, resume = int(info[i, 0]), int(info[i, 1])
        got = lists[i, :count].astype(np.int64)
        assert count <= LIST_CAP and resume <= c.size + 600
        assert (np.diff(got) > 0).all() and (count == 0 or got[-1] < c.size)


def _chase_backend(backend):
    """The persistent one-wave-per-chunk kernels, whatever the batch size, built WITH the index (-DNVCOMP_LZ_INDEX=1: the
    product leaves it out, csrc/api/lz4_api.hip says why)."""
    from conftest import Backend, emu_path_library, gpu_path_library

    lib = (emu_path_library if backend.name == "emu" else gpu_path_library)("index")
    return Backend(backend.name, lib, backend.dev)


def test_lz4_decode_through_index_and_chase(backend, oracle):
    """Streams whose index ends early (a long literal run in the middle, runs the walk cannot size, the last bytes of every
    chunk) are decoded by index and chase in turn: bit-exact, and the same with a temp buffer too small for any index."""
    rng = np.random.RandomState(7)
    base = datasets.text(70000, 2)
    noise = rng.randint(0, 256, 65536).astype(np.uint8)
    raws = [
        base[:65536],
        np.concatenate([base[:30000], noise[:8000], base[:20000]]),   # index, long literals (chase), nothing more indexed
        np.concatenate([base[:20000], np.zeros(30000, dtype=np.uint8), base[:15000]]),  # a match of 30 000 bytes in the middle
        np.concatenate([noise[:300], base[:40000]]),                  # a literal run of 300 bytes first
        datasets.table_rows(65536, 4),
        noise[:65000],
    ]
    b = _chase_backend(backend)
    codec = b.codec("LZ4")
    assert codec.decompress_temp_size(8, 65536) > 8 * 40000, "this build asks for room for the index"
    for level in (0, 12):
        comp = [oracle.ref_lz4_compress(r, level) for r in raws]
        outs, actual, status = codec.decompress(comp, [r.size for r in raws])
        assert (status == 0).all() and actual.tolist() == [r.size for r in raws]
        for o, r in zip(outs, raws):
            assert np.array_equal(o, r)
    # a corrupt stream on the indexed path: flipped bytes in the middle must fail or decode inside the slot, never hang
    bad = comp[0].copy()
    bad[5000:5040] ^= 0x5a
    outs, actual, status = codec.decompress([bad, comp[1]], [raws[0].size, raws[1].size])
    assert status[1] == 0 and np.array_equal(outs[1], raws[1])
