#!/usr/bin/env python3
"""Generate tests/golden/own_*.bin: streams of this library's OWN formats (Cascaded, Bitcomp, ANS),
written by the CPU models in oracle/, for the first 64 KiB chunk of the reference's two fixture files.
They pin the stream layouts across rounds: the HIP compressors must reproduce them byte for byte and
both decoders must invert them. The original chunks are not stored again: tests recover them from the
committed liblz4-HC golden streams (tests/golden/manifest.json). Runs anywhere oracle/ builds."""
import hashlib
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import oracle_py as O  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")

CONFIGS = {
    "cascaded_uint_2_1_1": ("Cascaded", (4096, 5, 2, 1, 1), lambda c: O.cascaded_compress(c, 4096, 5, 2, 1, 1)),
    "bitcomp_a0_uchar": ("Bitcomp", (0, 1), lambda c: O.bitcomp_compress(c, 0, 1)),
    "bitcomp_a1_uint": ("Bitcomp", (1, 5), lambda c: O.bitcomp_compress(c, 1, 4)),
    "ans": ("ANS", (0,), O.ans_compress),
}


def main():
    base = json.load(open(os.path.join(OUT, "manifest.json")))
    manifest = {"note": "own-format golden streams written by oracle/*_ref.c (scripts/make_golden_own.py)", "streams": []}
    for fname, entry in base["files"].items():
        rec = entry["chunks"][0]
        comp = np.fromfile(os.path.join(OUT, rec["streams"]["lz4_hc12"]["file"]), dtype=np.uint8)
        rc, chunk = O.lz4_decompress(comp, rec["bytes"])
        assert rc == 0 and hashlib.sha256(chunk.tobytes()).hexdigest() == rec["sha256"]
        for key, (fmt, opts, enc) in CONFIGS.items():
            stream = enc(chunk)
            name = f"own_{fname.split('.')[0]}_0_{key}.bin"
            stream.tofile(os.path.join(OUT, name))
            manifest["streams"].append({"file": name, "format": fmt, "opts": list(opts), "source": rec["streams"]["lz4_hc12"]["file"],
                                        "bytes": rec["bytes"], "sha256": rec["sha256"], "stream_bytes": int(stream.size),
                                        "stream_sha256": hashlib.sha256(stream.tobytes()).hexdigest()})
    json.dump(manifest, open(os.path.join(OUT, "own_manifest.json"), "w"), indent=1)
    print({s["file"]: s["stream_bytes"] for s in manifest["streams"]})


if __name__ == "__main__":
    main()
