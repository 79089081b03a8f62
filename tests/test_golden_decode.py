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
def test_decode_golden(backend, kind, fmt):
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
