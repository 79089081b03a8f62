#!/usr/bin/env python3
"""Golden DEFLATE / gzip vectors: the chunks of the reference's fixture files that tests/golden already holds (as liblz4
streams: tests/golden/manifest.json) written by zlib the way the reference's examples write them --
examples/deflate_cpu_compression.cu:82-104 (deflateInit2(9, -15)), :69-81 (compress2 minus its wrapper), Z_FIXED and
level 1 for block-kind coverage, and examples/gzip_gpu_decompression.cu:57-81 (deflateInit2(9, 15 | 16)).
Run once (zlib of this container); the streams and tests/golden/deflate_manifest.json are committed."""
import hashlib
import json
import os
import sys
import zlib

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def stream(data, level, wbits, strategy=zlib.Z_DEFAULT_STRATEGY):
    o = zlib.compressobj(level, zlib.DEFLATED, wbits, 8, strategy)
    return o.compress(data) + o.flush()


def main():
    from oracle import oracle_py as oracle

    oracle.build()
    manifest = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    out = {"zlib": zlib.ZLIB_VERSION, "streams": []}
    for name, entry in manifest["files"].items():
        for i, rec in enumerate(entry["chunks"]):
            lz4 = np.fromfile(os.path.join(GOLDEN, rec["streams"]["lz4_hc12"]["file"]), dtype=np.uint8)
            rc, chunk = oracle.lz4_decompress(lz4, rec["bytes"])
            assert rc == 0 and hashlib.sha256(chunk.tobytes()).hexdigest() == rec["sha256"]
            data = chunk.tobytes()
            kinds = {
                "deflate_l9": ("Deflate", stream(data, 9, -15)),
                "deflate_l1": ("Deflate", stream(data, 1, -15)),
                "deflate_fixed": ("Deflate", stream(data, 9, -15, zlib.Z_FIXED)),
                "deflate_compress2": ("Deflate", zlib.compress(data, 9)[2:-4]),
                "gzip_l9": ("Gzip", stream(data, 9, 15 | 16)),
            }
            stem = os.path.splitext(os.path.basename(name))[0]
            for kind, (fmt, blob) in kinds.items():
                fn = f"{stem}_{i}_{kind}.bin"
                open(os.path.join(GOLDEN, fn), "wb").write(blob)
                out["streams"].append({"file": fn, "format": fmt, "kind": kind, "bytes": rec["bytes"], "sha256": rec["sha256"],
                                       "stream_bytes": len(blob), "stream_sha256": hashlib.sha256(blob).hexdigest()})
    json.dump(out, open(os.path.join(GOLDEN, "deflate_manifest.json"), "w"), indent=1)
    print(len(out["streams"]), "streams,", sum(s["stream_bytes"] for s in out["streams"]), "bytes")


if __name__ == "__main__":
    main()
