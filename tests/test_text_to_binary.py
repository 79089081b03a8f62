"""benchmarks/text_to_binary.py (ours) against the output of the reference's own script on the reference's own fixture
(tests/golden/ExampleFloatData_col*_float.bin, scripts/make_golden_columns.py), and -- when the reference tree is
present, i.e. in the build container -- against the source CSV directly."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")
REF_CSV = "/root/reference/benchmarks/ExampleFloatData.csv"


def test_fixtures_are_what_the_manifest_says():
    man = json.load(open(os.path.join(GOLD, "columns_manifest.json")))
    assert len(man) == 3
    for name, rec in man.items():
        blob = open(os.path.join(GOLD, name), "rb").read()
        assert len(blob) == rec["bytes"] == 4001 * 4 and hashlib.sha256(blob).hexdigest() == rec["sha256"]
    x = np.fromfile(os.path.join(GOLD, "ExampleFloatData_col0_float.bin"), dtype=np.float32)
    assert x[0] == 0 and abs(x[1] - 0.01) < 1e-7 and abs(x[-1] - 40.0) < 1e-4  # the 0.01-step ramp of column 0


def test_round_trip_through_our_tool(tmp_path):
    """CSV written from the fixtures -> our tool -> the same bytes (float32 text with 9 significant digits is exact)."""
    cols = [np.fromfile(os.path.join(GOLD, f"ExampleFloatData_col{c}_float.bin"), dtype=np.float32) for c in range(3)]
    csv = tmp_path / "t.csv"
    with open(csv, "w") as f:
        for row in zip(*cols):
            f.write(",".join(np.format_float_positional(v, unique=True) for v in row) + "\n")
    for c in range(3):
        out = tmp_path / f"c{c}.bin"
        r = subprocess.run([sys.executable, os.path.join(REPO, "benchmarks", "text_to_binary.py"), str(csv), str(c), "float", str(out)],
                           check=True, capture_output=True, text=True)
        assert "Wrote 4001 floats" in r.stdout
        assert open(out, "rb").read() == cols[c].tobytes()
    # other types and a delimiter
    tab = tmp_path / "t.txt"
    tab.write_text("1|-5|2.5|abc\n2|70000|1e-3|de\n")
    sys.path.insert(0, os.path.join(REPO, "benchmarks"))
    import text_to_binary as t2b

    assert t2b.convert(str(tab), 1, "int", str(tmp_path / "i.bin"), "|") == 2
    assert np.fromfile(tmp_path / "i.bin", dtype=np.int32).tolist() == [-5, 70000]
    t2b.convert(str(tab), 0, "long", str(tmp_path / "l.bin"), "|")
    assert np.fromfile(tmp_path / "l.bin", dtype=np.int64).tolist() == [1, 2]
    t2b.convert(str(tab), 2, "double", str(tmp_path / "d.bin"), "|")
    assert np.fromfile(tmp_path / "d.bin", dtype=np.float64).tolist() == [2.5, 1e-3]


@pytest.mark.skipif(not os.path.exists(REF_CSV), reason="the reference tree is only in the build container")
def test_same_bytes_as_the_reference_script(tmp_path):
    for c in range(3):
        out = tmp_path / f"c{c}.bin"
        subprocess.run([sys.executable, os.path.join(REPO, "benchmarks", "text_to_binary.py"), REF_CSV, str(c), "float", str(out)],
                       check=True, capture_output=True)
        assert open(out, "rb").read() == open(os.path.join(GOLD, f"ExampleFloatData_col{c}_float.bin"), "rb").read()
