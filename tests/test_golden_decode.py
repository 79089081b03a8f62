"""HIP decoders (and their host emulation) on the committed golden vectors: streams written
by liblz4 / libsnappy from the reference's own fixture files (tests/golden/manifest.json)."""
import hashlib
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MANIFEST = json.load(open(os.path.join(GOLDEN, "manifest.json")))


@pytest.mark.parametrize("kind,fmt", [("lz4_default", "LZ4"), ("lz4_hc12", "LZ4"), ("snappy", "Snappy")])
def test_decode_golden(backend, lz_path, kind, fmt):
    comp, recs = [], []
    for entry in MANIFEST["files"].values():
        for rec in entry["chunks"]:
            comp.append(np.fromfile(os.path.join(GOLDEN, rec["streams"][kind]["file"]), dtype=np.uint8))
            recs.append(rec)
    codec = backend.codec(fmt)
    outs, actual, status = codec.decompress(comp, [r["bytes"] for r in recs])
    assert (status == 0).all() and actual.tolist() == [r["bytes"] for r in recs]
    for o, r in zip(outs, recs):
        assert hashlib.sha256(o.tobytes()).hexdigest() == r["sha256"]
    assert codec.get_decompress_size(comp).tolist() == [r["bytes"] for r in recs]


OWN = json.load(open(os.path.join(GOLDEN, "own_manifest.json")))


@pytest.mark.parametrize("rec", OWN["streams"], ids=[s["file"] for s in OWN["streams"]])
def test_own_format_golden(backend, oracle, rec):
    """Cascaded / Bitcomp / ANS streams committed by scripts/make_golden_own.py: the HIP compressor must
    reproduce them byte for byte and the HIP decompressor must invert them (layout pinned across rounds)."""
    stream = np.fromfile(os.path.join(GOLDEN, rec["file"]), dtype=np.uint8)
    assert hashlib.sha256(stream.tobytes()).hexdigest() == rec["stream_sha256"]
    rc, chunk = oracle.lz4_decompress(np.fromfile(os.path.join(GOLDEN, rec["source"]), dtype=np.uint8), rec["bytes"])
    assert rc == 0 and hashlib.sha256(chunk.tobytes()).hexdigest() == rec["sha256"]
    codec = backend.codec(rec["format"], tuple(rec["opts"]))
    (made,) = codec.compress([chunk])
    assert np.array_equal(made, stream), "compressed bytes differ from the committed golden stream"
    outs, actual, status = codec.decompress([stream], [rec["bytes"]], comp_align=8, out_align=8)
    assert status[0] == 0 and actual[0] == rec["bytes"] and np.array_equal(outs[0], chunk)


DEFLATE = json.load(open(os.path.join(GOLDEN, "deflate_manifest.json")))


@pytest.mark.parametrize("fmt", ["Deflate", "Gzip"])
def test_deflate_golden(backend, fmt):
    """DEFLATE / gzip streams committed by scripts/make_golden_deflate.py: zlib's output for the reference's fixture
    chunks, written the way examples/deflate_cpu_compression.cu and gzip_gpu_decompression.cu write them."""
    recs = [s for s in DEFLATE["streams"] if s["format"] == fmt]
    if backend.name != "gpu":
        recs = recs[: len(recs) // 2]  # every kind of the first file's chunks: the emulator decodes ~50 KB/s
    comp = [np.fromfile(os.path.join(GOLDEN, r["file"]), dtype=np.uint8) for r in recs]
    for c, r in zip(comp, recs):
        assert hashlib.sha256(c.tobytes()).hexdigest() == r["stream_sha256"]
    codec = backend.codec(fmt)
    outs, actual, status = codec.decompress(comp, [r["bytes"] for r in recs])
    assert (status == 0).all() and actual.tolist() == [r["bytes"] for r in recs]
    for o, r in zip(outs, recs):
        assert hashlib.sha256(o.tobytes()).hexdigest() == r["sha256"]
    few = slice(0, None if backend.name == "gpu" else 4)  # the size query decodes symbol by symbol: slow under emulation
    assert codec.get_decompress_size(comp[few]).tolist() == [r["bytes"] for r in recs[few]]
