"""Synthetic workloads for tests and bench.py (numpy, fixed seeds, no files needed).

The reference's harness reads user files (benchmarks/benchmark_template_chunked.cuh:313-356)
and ships only two small fixtures plus three generators:
  * ``gen_data(max_byte, size)``: uniform bytes in [0, max_byte]
    (benchmarks/benchmark_common.h:158-175); max_byte=3 is the Snappy workload
    (benchmarks/benchmark_snappy_synth.cpp:57-61), 255 and 0 the LZ4 ones
    (benchmarks/benchmark_lz4_synth.cpp:64-72);
  * 1,000,000 uniform bytes (examples/low_level_quickstart_example.cpp:163-176).
Silesia is not available offline, so the "Silesia-style mix" BASELINE.json asks
for is synthesised here from classes that span the same range of match/literal
statistics: natural-language-like text, '|'-separated table rows, decimal CSV,
float32 / int32 columns, low-cardinality bytes, zeros and incompressible noise.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Sequence

import numpy as np

CHUNK = 1 << 16  # default chunk size of every reference benchmark (-p 65536)


def gen_data(max_byte: int, size: int, seed: int = 0) -> np.ndarray:
    """Uniform bytes in [0, max_byte] (shape of benchmarks/benchmark_common.h:158-175)."""
    rng = np.random.RandomState(seed)
    return rng.randint(0, max_byte + 1, size=size, dtype=np.int64).astype(np.uint8)


def zeros(size: int, seed: int = 0) -> np.ndarray:
    return np.zeros(size, dtype=np.uint8)


def noise(size: int, seed: int = 0) -> np.ndarray:
    return gen_data(255, size, seed)


def lowcard(size: int, seed: int = 0) -> np.ndarray:
    return gen_data(3, size, seed)


def _vocabulary(rng: np.random.RandomState, words: int, min_len: int, max_len: int, alphabet: bytes) -> tuple:
    letters = np.frombuffer(alphabet, dtype=np.uint8)
    # skewed letter frequencies, like natural language
    p = 1.0 / np.arange(1, len(letters) + 1) ** 0.8
    p /= p.sum()
    lens = rng.randint(min_len, max_len + 1, size=words)
    table = letters[rng.choice(len(letters), size=(words, max_len + 1), p=p)]
    return table, lens


def _emit_words(table: np.ndarray, lens: np.ndarray, ids: np.ndarray, sep: np.ndarray) -> np.ndarray:
    """Concatenate table[id, :lens[id]] + sep[i] for every id, vectorised."""
    rows = table[ids].copy()
    l = lens[ids]
    rows[np.arange(len(ids)), l] = sep
    mask = np.arange(table.shape[1])[None, :] <= l[:, None]
    return rows[mask]


def text(size: int, seed: int = 0) -> np.ndarray:
    """Zipf-distributed words from a 8192-word vocabulary, sentence punctuation."""
    rng = np.random.RandomState(seed + 101)
    table, lens = _vocabulary(rng, 8192, 2, 11, b"etaoinshrdlcumwfgypbvkjxqz")
    out: List[np.ndarray] = []
    have = 0
    rank_p = 1.0 / np.arange(1, 8193) ** 1.05
    rank_p /= rank_p.sum()
    while have < size:
        n = max(1024, (size - have) // 5 + 16)
        ids = rng.choice(8192, size=n, p=rank_p)
        sep = np.full(n, ord(" "), dtype=np.uint8)
        r = rng.rand(n)
        sep[r < 0.08] = ord(",")
        sep[r < 0.05] = ord(".")
        sep[r < 0.01] = ord("\n")
        piece = _emit_words(table, lens, ids, sep)
        out.append(piece)
        have += piece.size
    return np.concatenate(out)[:size]


def table_rows(size: int, seed: int = 0) -> np.ndarray:
    """'|'-separated rows in the style of a TPC-H lineitem dump (the shape of the
    reference's benchmarks/ExampleTable.txt): keys, quantities, decimals, flags,
    dates, ship modes and a free-text comment."""
    rng = np.random.RandomState(seed + 202)
    table, lens = _vocabulary(rng, 512, 3, 9, b"etaoinshrdlcumwfgypbvkjxqz")
    modes = [b"TRUCK", b"MAIL", b"SHIP", b"AIR", b"RAIL", b"FOB", b"REG AIR"]
    instr = [b"DELIVER IN PERSON", b"COLLECT COD", b"NONE", b"TAKE BACK RETURN"]
    rows: List[bytes] = []
    have = 0
    key = 1
    while have < size:
        n_items = rng.randint(1, 8)
        for line in range(1, n_items + 1):
            qty = rng.randint(1, 51)
            price = rng.randint(90000, 10500000) / 100.0
            y, m, d = rng.randint(1992, 1999), rng.randint(1, 13), rng.randint(1, 29)
            words = rng.randint(0, 512, size=rng.randint(3, 9))
            comment = b" ".join(bytes(table[w, : lens[w]]) for w in words)
            row = b"|".join([
                str(key).encode(), str(rng.randint(1, 200001)).encode(), str(rng.randint(1, 10001)).encode(),
                str(line).encode(), str(qty).encode(), ("%.2f" % price).encode(),
                ("0.%02d" % rng.randint(0, 11)).encode(), ("0.%02d" % rng.randint(0, 9)).encode(),
                rng.choice([b"N", b"R", b"A"]), rng.choice([b"O", b"F"]),
                ("%04d-%02d-%02d" % (y, m, d)).encode(), ("%04d-%02d-%02d" % (y, m, min(28, d + 3))).encode(),
                ("%04d-%02d-%02d" % (y, min(12, m + 1), d)).encode(),
                instr[rng.randint(0, 4)], modes[rng.randint(0, 7)], comment,
            ]) + b"|\n"
            rows.append(row)
            have += len(row)
        key += rng.randint(1, 4)
    return np.frombuffer(b"".join(rows), dtype=np.uint8)[:size].copy()


def float_csv(size: int, seed: int = 0) -> np.ndarray:
    """Three comma-separated decimal columns per line (the shape of the
    reference's benchmarks/ExampleFloatData.csv)."""
    rng = np.random.RandomState(seed + 303)
    n = size // 24 + 8
    a = rng.rand(n)
    b = np.cumsum(rng.randn(n)) * 0.01 + 5.0
    c = rng.randint(0, 1000, size=n) / 8.0
    lines = "".join("%.8f,%.6f,%.3f\n" % t for t in zip(a, b, c))
    return np.frombuffer(lines.encode(), dtype=np.uint8)[:size].copy()


def float32_column(size: int, seed: int = 0) -> np.ndarray:
    """Smooth sensor-like signal quantised to 1/64, stored as float32."""
    rng = np.random.RandomState(seed + 404)
    n = size // 4 + 1
    x = np.cumsum(rng.randn(n).astype(np.float32)) * 0.25
    x = np.round(x * 64.0) / 64.0
    return x.astype(np.float32).view(np.uint8)[:size].copy()


def float_columns(size: int, seed: int = 0) -> np.ndarray:
    """The shape of the reference's ExampleFloatData.csv columns after text_to_binary.py (BASELINE.json configs[3],
    "int32 columnar floats"): 4001-row blocks of a 0.01-step ramp, a slowly varying signal with 9 significant digits
    and a second, flatter one, as float32; the blocks repeat with a drifting phase like `-x` duplication would not --
    so that chunks differ. No two neighbours are equal, the deltas are small and smooth."""
    rng = np.random.RandomState(seed + 606)
    rows = 4001
    blocks = size // (4 * rows) + 1
    out = np.empty(blocks * rows, dtype=np.float32)
    for b in range(blocks):
        t = np.arange(rows, dtype=np.float64) * 0.01
        kind = b % 3
        if kind == 0:
            col = t + 40.0 * (b // 3)
        elif kind == 1:
            ph = rng.rand() * 6.28
            col = 0.5 + 0.07 * np.sin(0.31 * t + ph) + 0.011 * np.sin(2.7 * t + 2 * ph) + 1e-5 * rng.randn(rows)
        else:
            ph = rng.rand() * 6.28
            col = 2.29 + 0.004 * np.cos(0.11 * t + ph) + 2e-6 * np.cumsum(rng.randn(rows))
        out[b * rows: (b + 1) * rows] = col.astype(np.float32)
    return out.view(np.uint8)[:size].copy()


def example_float_columns(size: int, seed: int = 0) -> np.ndarray:
    """The REAL BASELINE.json configs[3] input: the three columns of the reference's benchmarks/ExampleFloatData.csv
    after text_to_binary.py (float32, 4001 values each; fixtures tests/golden/ExampleFloatData_col*_float.bin, made by
    the reference's own script: scripts/make_golden_columns.py), repeated column after column to `size` bytes the way
    the harness's `-x` duplication multiplies a file. `seed` rotates the starting column."""
    import os

    gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    cols = [np.fromfile(os.path.join(gold, f"ExampleFloatData_col{c}_float.bin"), dtype=np.uint8) for c in range(3)]
    cols = cols[seed % 3:] + cols[:seed % 3]
    unit = np.concatenate(cols)
    return np.tile(unit, size // unit.size + 1)[:size].copy()


def mortgage_col0_like(size: int, seed: int = 0) -> np.ndarray:
    """The shape of the one dataset the reference publishes LZ4 numbers for (doc/Benchmarks.md:88-95: Mortgage 2009Q2,
    column 0 as int64, 329 055 928 bytes, LZ4 ratio 38.89): a sorted key column of a fact table -- 12-digit loan ids,
    each repeated once per monthly record (a few dozen rows), ids growing by irregular gaps. liblz4 compresses it
    39x: one 8-byte-period match per run of equal ids."""
    rng = np.random.RandomState(seed + 909)
    n = size // 8 + 1
    runs = rng.randint(16, 86, size=n // 16 + 2)  # rows per loan
    ids = 100000000000 + np.cumsum(rng.randint(1, 5000, size=runs.size).astype(np.int64))
    col = np.repeat(ids, runs)[:n]
    return col.view(np.uint8)[:size].copy()


def int32_column(size: int, seed: int = 0) -> np.ndarray:
    """Sorted keys with runs (low-cardinality dimension column), int32."""
    rng = np.random.RandomState(seed + 505)
    n = size // 4 + 1
    steps = (rng.rand(n) < 0.02).astype(np.int32) * rng.randint(1, 5, size=n).astype(np.int32)
    return (np.cumsum(steps, dtype=np.int64) + 1000).astype(np.int32).view(np.uint8)[:size].copy()


CLASSES: Dict[str, Callable[[int, int], np.ndarray]] = {
    "example_float_columns": example_float_columns,
    "mortgage_col0_like": mortgage_col0_like,
    "text": text,
    "table": table_rows,
    "float_csv": float_csv,
    "float32": float32_column,
    "int32": int32_column,
    "float_columns": float_columns,
    "lowcard": lowcard,
    "zeros": zeros,
    "noise": noise,
}

# Byte shares of the Silesia-style mix (text/markup-heavy, some binary, a little
# incompressible and a little trivially compressible data).
SILESIA_STYLE_MIX = (
    ("text", 0.34),
    ("table", 0.22),
    ("float_csv", 0.12),
    ("float32", 0.10),
    ("int32", 0.08),
    ("lowcard", 0.06),
    ("noise", 0.05),
    ("zeros", 0.03),
)


def silesia_style(size: int, seed: int = 0, chunk: int = CHUNK) -> np.ndarray:
    """`size` bytes of the mix, interleaved chunk by chunk so neighbouring
    wavefronts see different statistics (SURVEY.md 8(d) class G)."""
    n_chunks = (size + chunk - 1) // chunk
    rng = np.random.RandomState(seed + 606)
    names = [n for n, _ in SILESIA_STYLE_MIX]
    shares = np.array([s for _, s in SILESIA_STYLE_MIX])
    counts = np.floor(shares * n_chunks).astype(int)
    counts[0] += n_chunks - counts.sum()
    streams = {n: CLASSES[n](int(c) * chunk, seed) if c else np.zeros(0, np.uint8) for n, c in zip(names, counts)}
    order = np.repeat(np.arange(len(names)), counts)
    rng.shuffle(order)
    used = dict.fromkeys(names, 0)
    out = np.empty(n_chunks * chunk, dtype=np.uint8)
    for i, k in enumerate(order):
        nme = names[k]
        out[i * chunk: (i + 1) * chunk] = streams[nme][used[nme]: used[nme] + chunk]
        used[nme] += chunk
    return out[:size]


def split_chunks(data: np.ndarray, chunk: int = CHUNK) -> List[np.ndarray]:
    """Cut a buffer into <= chunk-byte pieces, the last one short
    (benchmarks/benchmark_template_chunked.cuh:323-333, examples/util.h:63-79)."""
    data = np.asarray(data).view(np.uint8).reshape(-1)
    return [data[i: i + chunk] for i in range(0, data.size, chunk)]
