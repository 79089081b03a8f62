"""Pins the CPU oracle (oracle/*.c) before anything trusts it:
  * against the committed golden vectors (tests/golden/, made by scripts/make_golden.py from
    the reference's own fixture files with liblz4 1.9.3 / snappy 1.1.8);
  * against liblz4 / libsnappy themselves where the container has them (oracle/_ref);
  * against the reference's whole-file known answers when
    /root/reference is present;
  * on corrupted streams: whatever the library decoders accept, the oracle accepts identically."""
import hashlib
import json
import os

import numpy as np
import pytest

from nvcomp_amd import datasets

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MANIFEST = json.load(open(os.path.join(GOLDEN, "manifest.json")))


def golden_cases():
    for fname, entry in MANIFEST["files"].items():
        for rec in entry["chunks"]:
            for kind, st in rec["streams"].items():
                yield pytest.param(fname, rec, kind, st, id=f"{fname}-{rec['index']}-{kind}")


@pytest.mark.parametrize("fname,rec,kind,st", list(golden_cases()))
def test_oracle_decodes_golden_vectors(oracle, fname, rec, kind, st):
    comp = np.fromfile(os.path.join(GOLDEN, st["file"]), dtype=np.uint8)
    assert comp.size == st["bytes"]
    if kind.startswith("lz4"):
        rc, out = oracle.lz4_decompress(comp, rec["bytes"])
        assert oracle.lz4_decompressed_size(comp) == rec["bytes"]
    else:
        rc, out = oracle.snappy_decompress(comp, rec["bytes"])
        assert oracle.snappy_decompressed_size(comp) == (0, rec["bytes"])
    assert rc == 0 and out.size == rec["bytes"]
    assert hashlib.sha256(out.tobytes()).hexdigest() == rec["sha256"]


def test_reference_known_answers(oracle):
    """Whole-file totals of BASELINE.md section 2; needs the reference tree and liblz4/snappy."""
    ref = "/root/reference/benchmarks"
    if not (os.path.isdir(ref) and oracle.have_ref()):
        pytest.skip("reference tree or liblz4/snappy not available here")
    for fname, entry in MANIFEST["files"].items():
        raw = np.fromfile(os.path.join(ref, fname), dtype=np.uint8)
        assert hashlib.md5(raw.tobytes()).hexdigest() == entry["md5"]
        chunks = datasets.split_chunks(raw)
        assert len(chunks) == entry["num_chunks"]
        tot = {"lz4_default": 0, "lz4_hc12": 0, "snappy": 0}
        for c in chunks:
            for kind, comp in (("lz4_default", oracle.ref_lz4_compress(c)), ("lz4_hc12", oracle.ref_lz4_compress(c, 12)),
                               ("snappy", oracle.ref_snappy_compress(c))):
                tot[kind] += comp.size
                dec = oracle.lz4_decompress if kind.startswith("lz4") else oracle.snappy_decompress
                rc, out = dec(comp, c.size)
                assert rc == 0 and np.array_equal(out, c)
        assert tot == entry["total_compressed_bytes"]
    assert MANIFEST["files"]["ExampleFloatData.csv"]["total_compressed_bytes"] == {
        "lz4_default": 75314, "lz4_hc12": 64849, "snappy": 75487}


@pytest.mark.parametrize("name", sorted(datasets.CLASSES))
def test_oracle_agrees_with_libraries(oracle, name):
    if not oracle.have_ref():
        pytest.skip("liblz4/snappy not available here")
    data = datasets.CLASSES[name](65536 + 1000, 3)
    for c in datasets.split_chunks(data):
        for comp in (oracle.ref_lz4_compress(c), oracle.ref_lz4_compress(c, 9), oracle.lz4_compress(c)):
            a = oracle.lz4_decompress(comp, c.size)
            b = oracle.ref_lz4_decompress(comp, c.size)
            assert a[0] == 0 and b[0] == 0 and np.array_equal(a[1], c) and np.array_equal(b[1], c)
        for comp in (oracle.ref_snappy_compress(c), oracle.snappy_compress(c)):
            a = oracle.snappy_decompress(comp, c.size)
            b = oracle.ref_snappy_decompress(comp, c.size)
            assert a[0] == 0 and b[0] == 0 and np.array_equal(a[1], c) and np.array_equal(b[1], c)
    assert oracle.lz4_bound(65536) == 65536 + 65536 // 255 + 16
    assert oracle.snappy_bound(65536) == 32 + 65536 + 65536 // 6


def test_oracle_on_corrupt_streams(oracle):
    if not oracle.have_ref():
        pytest.skip("liblz4/snappy not available here")
    rng = np.random.RandomState(11)
    base = datasets.text(6000, 2)
    lz = oracle.ref_lz4_compress(base)
    sn = oracle.ref_snappy_compress(base)
    accepted = 0
    for trial in range(400):
        for comp, ours, theirs in ((lz, oracle.lz4_decompress, oracle.ref_lz4_decompress),
                                   (sn, oracle.snappy_decompress, oracle.ref_snappy_decompress)):
            b = comp.copy()
            k = rng.randint(0, 3)
            if k == 0:
                b[rng.randint(0, b.size)] = rng.randint(0, 256)
            elif k == 1:
                b = b[: rng.randint(1, b.size)]
            else:
                b[rng.randint(0, b.size)] ^= 1 << rng.randint(0, 8)
            rc_t, out_t = theirs(b, base.size)
            rc_o, out_o = ours(b, base.size)
            if rc_t == 0:  # the library decoder accepts: the oracle must agree byte for byte
                accepted += 1
                assert rc_o == 0 and np.array_equal(out_o, out_t)
    assert accepted > 0
