"""Cascaded codec: HIP path (or its host emulation) vs oracle/cascaded_ref.c.

The reference documents the scheme but not the bitstream (doc/cascaded_overview.md),
so parity is pinned to this library's own container: compressed bytes must be
IDENTICAL to the CPU model's, decompression must invert both, over every element
type and RLE/delta/bit-packing combination (benchmarks/benchmark_cascaded_chunked.cu:
35-36 defaults {4096, type, 2, 1, 1}; benchmark_all_algorithms.sh:5-8 uses 1/0/1, 0/0/1)."""
import numpy as np
import pytest

from nvcomp_amd import datasets
from nvcomp_amd._lib import NvcompStatus

WIDTH = [1, 1, 2, 2, 4, 4, 8, 8]


def roundtrip(backend, oracle, chunks, opts):
    sub, typ, r, d, bp = opts
    codec = backend.codec("Cascaded", opts)
    comp = codec.compress(chunks, in_align=8)
    for i, (cc, c) in enumerate(zip(comp, chunks)):
        ref = oracle.cascaded_compress(c, sub, typ, r, d, bp)
        assert cc.size == ref.size and np.array_equal(cc, ref), f"chunk {i}: compressed bytes differ from the CPU model"
        rc, out = oracle.cascaded_decompress(cc, c.size)
        assert rc == 0 and np.array_equal(out, c)
    outs, actual, status = codec.decompress(comp, [c.size for c in chunks], comp_align=8, out_align=8)
    assert (status == NvcompStatus.Success).all(), status
    assert actual.tolist() == [c.size for c in chunks]
    for o, c in zip(outs, chunks):
        assert np.array_equal(o, c)
    sizes = codec.get_decompress_size(comp, comp_align=8)
    assert sizes.tolist() == [c.size for c in chunks]
    return sum(c.size for c in chunks) / max(1, sum(c.size for c in comp))


@pytest.mark.parametrize("typ", range(8))
def test_default_scheme_all_types(backend, oracle, typ):
    rng = np.random.RandomState(typ)
    w = WIDTH[typ]
    chunks = []
    for name in ("int32", "float32", "lowcard", "zeros", "noise"):
        n = int(rng.randint(1, 2500)) * 8
        chunks.append(datasets.CLASSES[name](n, typ))
    chunks.append(np.zeros(0, dtype=np.uint8))
    chunks.append(datasets.int32_column(65536, 3))
    roundtrip(backend, oracle, chunks, (4096, typ, 2, 1, 1))
    assert w in (1, 2, 4, 8)


@pytest.mark.parametrize("r,d,bp", [(0, 0, 1), (1, 0, 1), (0, 1, 1), (2, 2, 1), (3, 1, 1), (1, 2, 0), (2, 1, 0), (0, 0, 0),
                                     (7, 7, 1)])
def test_scheme_combinations(backend, oracle, r, d, bp):
    chunks = [datasets.int32_column(20000, r + d), datasets.float32_column(8192, 1), datasets.lowcard(5000 * 4, 2),
              datasets.noise(4096, 3)]
    for typ in (1, 4, 7):
        if r == 7 and typ == 1:
            # 7 RLE layers x 4096 one-byte elements need more LDS than one workgroup may hold
            with pytest.raises(RuntimeError, match="returned 11"):  # nvcompErrorNotSupported
                backend.codec("Cascaded", (4096, typ, r, d, bp)).compress(chunks)
            continue
        roundtrip(backend, oracle, chunks, (4096, typ, r, d, bp))


@pytest.mark.parametrize("typ", [1, 3, 4, 6])
@pytest.mark.parametrize("r,d", [(2, 1), (1, 0), (1, 1), (3, 2)])
def test_short_run_expansion(backend, oracle, typ, r, d):
    """Values with short runs (smooth float columns, BASELINE.json configs[3]): when every expanding layer's runs are
    at most 64 the decoder expands them straight from the packed run streams (casc::rle_expand_direct to HBM,
    rle_expand_inplace for inner layers); one longer run anywhere sends the sub-chunk down the pool + marks path.
    Both sides of every boundary, every element width."""
    w = WIDTH[typ]
    rng = np.random.RandomState(100 * typ + 10 * r + d)
    dt = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[w]

    def column(run_lengths, n_runs):
        runs = rng.choice(run_lengths, size=n_runs, p=None)
        vals = (np.cumsum(rng.randint(1, 7, size=n_runs)) % (1 << (8 * w - 1) if w < 8 else 1 << 62)).astype(dt)
        vals[1:][vals[1:] == vals[:-1]] += 1  # neighbours stay distinct: the second RLE layer finds nothing
        return np.repeat(vals, runs).view(np.uint8)

    chunks = [
        column([1, 1, 1, 1, 1, 1, 1, 2], 3000),        # bits 1, min 1
        column([1, 1, 1, 1, 2, 3, 4], 3000),            # bits 2
        column([1, 1, 1, 5, 8], 2000),                  # bits 3, max run exactly 8
        column([1, 1, 1, 1, 9], 2000),                  # bits 4
        column([1, 2, 64], 400),                        # bits 6, max run exactly 64
        column([1, 2, 65], 400),                        # 65 > 64: pool + marks path
        column([2, 3], 2000),                           # min 2
        column([1, 1, 1, 1, 1, 300], 500),              # one long run per few: general path
        np.repeat(column([1, 2], 1500).view(dt), 2).view(np.uint8),  # every value twice: the inner layer expands too
        datasets.float_columns(65536, typ),
    ]
    chunks = [c[: c.size // w * w] for c in chunks]
    roundtrip(backend, oracle, chunks, (4096, typ, r, d, 1))


@pytest.mark.parametrize("typ", [1, 3, 4, 6])
@pytest.mark.parametrize("r,d,bp", [(2, 1, 1), (1, 0, 1), (2, 0, 0), (3, 1, 1)])
def test_run_pool_choice(backend, oracle, typ, r, d, bp):
    """The compressor's first pass has room for one run per TWO elements in its two-byte pool; a layer with more runs takes a
    one-byte pool (start indices modulo 256), which in turn cannot hold a run of 256 elements or more (casc::rle_encode,
    compress_sub: the choice is tried, retried the other way when the input is still in memory, hinted from the chunk's
    previous sub-chunk). Sub-chunks on every side of both limits, in every order inside one chunk -- the hint is wrong at
    each change --, runs of 254 / 255 / 256 / 257 / 300 elements among a run per element, at the start, in the middle and at
    the end of a sub-chunk. The bytes must be the CPU model's whatever path produced them."""
    w = WIDTH[typ]
    rng = np.random.RandomState(1000 * typ + 100 * r + 10 * d + bp)
    dt = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[w]
    n = 4096 // w  # elements per sub-chunk

    def distinct(k):  # k elements, neighbours differ
        v = np.cumsum(rng.randint(1, 5, size=k)).astype(np.uint64)
        return (v % (1 << (8 * w - 1) if w < 8 else 1 << 62)).astype(dt)

    def crowded_with_run(run, where):  # a run per element, and ONE run of `run` elements
        rest = n - run
        a = {"start": 0, "middle": rest // 2, "end": rest}[where]
        v = distinct(rest + 1)
        return np.concatenate([v[:a], np.repeat(v[a: a + 1] + 1, run), v[a + 1: rest + 1]])[:n]

    def few_runs():  # fits the two-byte pool: runs of 2 .. 700
        parts, total = [], 0
        while total < n:
            k = int(rng.choice([2, 3, 50, 255, 256, 700]))
            parts.append(np.full(k, rng.randint(0, 200), dtype=dt))
            total += k
        return np.concatenate(parts)[:n]

    subs = [distinct(n), few_runs(), distinct(n)]
    for run in (254, 255, 256, 257, 300):
        for where in ("start", "middle", "end"):
            if run < n:
                subs += [crowded_with_run(run, where), distinct(n)]
    subs += [few_runs(), crowded_with_run(min(255, n // 2), "middle"), np.repeat(distinct(n // 2), 2), few_runs(), distinct(n - 7)]
    flat = np.concatenate(subs).astype(dt)
    per_chunk = 16 * n
    chunks = [flat[i: i + per_chunk].view(np.uint8) for i in range(0, flat.size, per_chunk)]
    roundtrip(backend, oracle, chunks, (4096, typ, r, d, bp))


def test_sub_chunk_sizes(backend, oracle):
    chunks = [datasets.int32_column(40000, 9), datasets.table_rows(12000, 4)]
    for sub in (256, 512, 1000, 4096, 8192, 16384):
        roundtrip(backend, oracle, chunks, (sub, 4, 2, 1, 1))
    # these need more than the 16 KiB/wave of the first decode pass -> large-LDS second pass
    roundtrip(backend, oracle, chunks, (8192, 0, 1, 1, 1))
    roundtrip(backend, oracle, chunks, (16384, 6, 2, 1, 1))


def test_later_passes_over_many_chunks(backend, oracle):
    """The passes behind the first are launched with the workgroups the card holds, not with a grid over the batch: a wave
    (small batches: a workgroup) takes every G-th chunk and finds its work in ONE load of up to 64 flag words
    (api/cascaded_api.hip: cascaded_{de,}compress_kernel<later>). Batches in which every third / every chunk / one chunk in
    seven needs the second or the third pass, of more chunks than those launches have waves (several chunks a wave, in the
    workgroup-per-chunk launch and in the wave-per-chunk launch), and -- on the card -- of more than 64 chunks a wave (the
    second round of flag words). Streams from different option sets share a decode batch, as the decoder allows."""
    few = backend.name == "emu"
    rng = np.random.RandomState(77)

    def runs_of(n, w, lengths):
        dt = {1: np.uint8, 4: np.uint32, 8: np.uint64}[w]
        k = n // w
        vals = np.cumsum(rng.randint(1, 9, size=k)).astype(np.uint64).astype(dt)
        return np.repeat(vals, rng.choice(lengths, size=k))[:k].view(np.uint8)

    # compress: (16384, 6, 2, 1, 1) has three passes (the worst case of its slice is beyond the middle one), (4096, 4, 2, 1, 1)
    # two; a run per element or two overflows the first pass's pools, long runs and smooth values do not
    kinds8 = [runs_of(16384, 8, [1, 1, 1, 2]), runs_of(16384, 8, [300, 500]), runs_of(16384, 8, [1]), datasets.noise(16384, 1)]
    kinds4 = [runs_of(8192, 4, [1, 1, 2]), runs_of(8192, 4, [40, 300]), datasets.float32_column(2048, 3), runs_of(8192, 4, [1])]
    n_comp = 150 if few else 2600  # (the card: 650 workgroups in the first pass, 512 / 1 024 resident behind it)
    for opts, kinds, pattern in (((16384, 6, 2, 1, 1), kinds8, (0, 1, 2)), ((16384, 6, 2, 1, 1), kinds8, (0, 0, 3, 0)),
                                 ((4096, 4, 2, 1, 1), kinds4, (0, 1, 2, 3, 1, 1, 1))):
        chunks = [kinds[pattern[i % len(pattern)]] for i in range(n_comp if opts[0] == 4096 or not few else 40)]
        codec = backend.codec("Cascaded", opts)
        comp = codec.compress(chunks, in_align=8)
        refs = [oracle.cascaded_compress(k, *opts) for k in kinds]
        for i, cc in enumerate(comp):
            ref = refs[pattern[i % len(pattern)]]
            assert cc.size == ref.size and np.array_equal(cc, ref), f"{opts}: chunk {i}: compressed bytes differ from the CPU model"
    # decode: streams of the first pass (smooth int32), of the second (long runs: pools and marks) and of the third (8 192
    # one-byte elements a sub-chunk; 16 384-byte sub-chunks of int64 runs)
    pool = []
    for opts, c in (((4096, 4, 2, 1, 1), datasets.int32_column(2048, 1)), ((4096, 4, 2, 1, 1), runs_of(8192, 4, [1, 2, 300])),
                    ((8192, 0, 1, 1, 1), runs_of(8192, 1, [1, 1, 3, 70])), ((16384, 6, 2, 1, 1), runs_of(16384, 8, [1, 2, 90])),
                    ((4096, 4, 2, 1, 1), datasets.float32_column(1024, 2))):
        cc = backend.codec("Cascaded", opts).compress([c], in_align=8)[0]
        rc, out = oracle.cascaded_decompress(cc, c.size)
        assert rc == 0 and np.array_equal(out, c)
        pool.append((cc, c))
    codec = backend.codec("Cascaded", (4096, 4, 2, 1, 1))
    sizes = (700,) if few else (700, 4096, 5000, 40000)
    for n in sizes:
        for pattern in ((0, 1, 2, 3, 4), (2, 3), (0, 0, 0, 4, 0, 0, 3), (1,)):
            if n == 40000 and pattern != (2, 3):
                continue
            picks = [pool[pattern[i % len(pattern)]] for i in range(n)]
            outs, actual, status = codec.decompress([cc for cc, _ in picks], [c.size for _, c in picks], comp_align=8, out_align=8)
            assert (status == NvcompStatus.Success).all(), (n, pattern, np.flatnonzero(status != NvcompStatus.Success)[:8])
            assert actual.tolist() == [c.size for _, c in picks]
            wrong = [i for i, (o, (_, c)) in enumerate(zip(outs, picks)) if not np.array_equal(o, c)]
            assert not wrong, f"{n} {pattern} {len(wrong)} {wrong[:80]}"


def test_benchmark_config_ratio(backend, oracle):
    """config 4 of BASELINE.json: int32 columnar data, default opts, 64 KiB user chunks."""
    data = datasets.int32_column(4 * 65536, 5)
    ratio = roundtrip(backend, oracle, datasets.split_chunks(data), (4096, 4, 2, 1, 1))
    assert ratio > 10


def test_corrupt_and_misaligned(backend, oracle):
    chunks = [datasets.int32_column(30000, 2)] * 6
    codec = backend.codec("Cascaded")
    comp = codec.compress(chunks)
    rng = np.random.RandomState(5)
    bad = []
    for i, c in enumerate(comp):
        b = c.copy()
        if i == 0:
            b = b[: b.size // 2]
        elif i == 1:
            b[0] ^= 0xFF
        elif i == 2:
            b[rng.randint(20, b.size)] ^= 0x10
        elif i == 3:
            b[8] ^= 0x40  # uncompressed size field
        bad.append(b)
    caps = [c.size for c in chunks]
    caps[5] -= 4
    outs, actual, status = codec.decompress(bad, caps, comp_align=8, out_align=8)
    for i, (b, cap) in enumerate(zip(bad, caps)):
        rc, ref = oracle.cascaded_decompress(b, cap)
        if rc == 0:
            assert status[i] == NvcompStatus.Success and np.array_equal(outs[i][: ref.size], ref)
        else:
            assert status[i] != NvcompStatus.Success and actual[i] == 0
    outs, actual, status = codec.decompress(comp, [c.size for c in chunks], comp_align=8, out_align=8, base_misalign=1)
    assert (status == NvcompStatus.ErrorAlignment).all()


def test_opts_validation(backend):
    import ctypes as C

    from nvcomp_amd._lib import CascadedOpts

    lib = backend.lib
    out = C.c_size_t(0)
    ok = CascadedOpts(4096, 4, 2, 1, 1)
    assert lib.nvcompBatchedCascadedCompressGetMaxOutputChunkSize(65536, ok, C.byref(out)) == 0
    assert out.value >= 65536
    for bad in (CascadedOpts(4096, 8, 2, 1, 1), CascadedOpts(4096, 4, 8, 1, 1), CascadedOpts(4096, 4, 2, 1, 2),
                CascadedOpts(100, 4, 2, 1, 1), CascadedOpts(4098, 4, 2, 1, 1)):
        assert lib.nvcompBatchedCascadedCompressGetMaxOutputChunkSize(65536, bad, C.byref(out)) == NvcompStatus.ErrorInvalidValue
