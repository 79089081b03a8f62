#!/usr/bin/env python3
"""tests/golden/ExampleFloatData_col{0,1,2}_float.bin: the three columns of the reference's ExampleFloatData.csv as
float32, produced by THE REFERENCE'S OWN benchmarks/text_to_binary.py, run here in the build container (the GPU box
has no /root/reference; 3 x 16 004 bytes travel as fixtures). They are the BASELINE.json configs[3] input
("Cascaded on int32 columnar floats": benchmark_cascaded_chunked over text_to_binary.py's output) and pin
benchmarks/text_to_binary.py (ours) byte for byte (tests/test_text_to_binary.py)."""
import hashlib
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/benchmarks"
OUT = os.path.join(REPO, "tests", "golden")


def main():
    manifest = {}
    for col in range(3):
        name = f"ExampleFloatData_col{col}_float.bin"
        out = os.path.join(OUT, name)
        subprocess.run([sys.executable, os.path.join(REF, "text_to_binary.py"), os.path.join(REF, "ExampleFloatData.csv"),
                        str(col), "float", out], check=True, cwd="/tmp", stdout=subprocess.DEVNULL)
        blob = open(out, "rb").read()
        manifest[name] = {"bytes": len(blob), "sha256": hashlib.sha256(blob).hexdigest(), "column": col, "dtype": "float32"}
    json.dump(manifest, open(os.path.join(OUT, "columns_manifest.json"), "w"), indent=1)
    print(json.dumps(manifest, indent=1))


if __name__ == "__main__":
    main()
