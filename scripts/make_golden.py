#!/usr/bin/env python3
"""Generate tests/golden/: golden vectors for the oracle and the HIP decoders.

Run in the build container (needs /root/reference for the two fixture files the
reference ships and oracle/_ref for liblz4 / libsnappy). For every 64 KiB chunk of
  benchmarks/ExampleFloatData.csv (2 chunks)   benchmarks/ExampleTable.txt (first 2 of 12 chunks)
it stores the streams written by LZ4_compress_default, LZ4_compress_HC(12) and
snappy::RawCompress, and in manifest.json the length + sha256 of the original chunk. The
manifest also records the whole-file known answers of BASELINE.md section 2 (total
compressed bytes per codec), which tests re-derive when the reference tree is present.
Splitting rule: examples/util.h:63-79 (each file cut independently, last chunk short)."""
import hashlib
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import oracle_py as O  # noqa: E402

REF = "/root/reference/benchmarks"
OUT = os.path.join(REPO, "tests", "golden")
CHUNK = 1 << 16


def main():
    assert O.have_ref(), "needs oracle/_ref (liblz4 + snappy)"
    os.makedirs(OUT, exist_ok=True)
    manifest = {"chunk_bytes": CHUNK, "liblz4": "1.9.3", "snappy": "1.1.8", "files": {}}
    for fname, keep in (("ExampleFloatData.csv", 2), ("ExampleTable.txt", 2)):
        raw = np.fromfile(os.path.join(REF, fname), dtype=np.uint8)
        chunks = [raw[i:i + CHUNK] for i in range(0, raw.size, CHUNK)]
        totals = {"lz4_default": 0, "lz4_hc12": 0, "snappy": 0}
        entry = {"bytes": int(raw.size), "md5": hashlib.md5(raw.tobytes()).hexdigest(), "num_chunks": len(chunks), "chunks": []}
        for i, c in enumerate(chunks):
            streams = {"lz4_default": O.ref_lz4_compress(c), "lz4_hc12": O.ref_lz4_compress(c, 12),
                       "snappy": O.ref_snappy_compress(c)}
            for k, v in streams.items():
                totals[k] += int(v.size)
            if i < keep:
                rec = {"index": i, "bytes": int(c.size), "sha256": hashlib.sha256(c.tobytes()).hexdigest(), "streams": {}}
                for k, v in streams.items():
                    name = f"{fname.split('.')[0]}_{i}_{k}.bin"
                    v.tofile(os.path.join(OUT, name))
                    rec["streams"][k] = {"file": name, "bytes": int(v.size)}
                entry["chunks"].append(rec)
        entry["total_compressed_bytes"] = totals
        manifest["files"][fname] = entry
    json.dump(manifest, open(os.path.join(OUT, "manifest.json"), "w"), indent=1)
    print(json.dumps({k: v["total_compressed_bytes"] for k, v in manifest["files"].items()}))


if __name__ == "__main__":
    main()
