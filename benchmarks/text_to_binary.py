#!/usr/bin/env python3
"""Column of a delimited text file -> raw binary array, the input preparation of the typed benchmarks.

Own implementation of the reference's tool of the same name and command line
(benchmarks/text_to_binary.py:81-118: `text_to_binary.py <file> <column> <int|long|float|double|string> <out> [delimiter]`,
default delimiter ","), e.g. the BASELINE.json configs[3] inputs:

    text_to_binary.py ExampleFloatData.csv 2 float ZValues.bin
    text_to_binary.py ExampleTable.txt 5 long Dates.bin '|'

tests/test_text_to_binary.py checks it byte for byte against files the reference's own script produced
(tests/golden/ExampleFloatData_col*_float.bin, made by scripts/make_golden.py in the build container)."""
import sys

import numpy as np

DTYPES = {"int": np.int32, "long": np.int64, "float": np.float32, "double": np.float64}


def convert(in_fname, column, datatype, out_fname, delimiter=","):
    """Returns the number of values written."""
    if datatype != "string" and datatype not in DTYPES:
        raise ValueError("datatype must be int, long, float, double or string")
    column = int(column)
    values = []
    with open(in_fname, "r") as f:
        for line in f:
            line = line.rstrip("\r\n")
            if not line.strip():
                continue
            fields = line.split(delimiter)
            if column >= len(fields):
                raise ValueError(f"line has {len(fields)} fields, column {column} asked for: {line!r}")
            values.append(fields[column])
    if datatype == "string":
        width = max((len(v) for v in values), default=1)
        arr = np.array(values, dtype=f"<U{width}")  # what numpy.genfromtxt(dtype=str).tofile() writes
    elif datatype in ("int", "long"):
        arr = np.array([int(v) for v in values], dtype=DTYPES[datatype])
    else:
        arr = np.array([float(v) for v in values], dtype=np.float64).astype(DTYPES[datatype])
    arr.tofile(out_fname)
    return arr.size


def main(argv):
    if len(argv) not in (5, 6):
        print("usage: text_to_binary.py <input text file> <column number> <int|long|float|double|string> "
              "<output binary file> [delimiter, default ',']")
        print("    text_to_binary.py ExampleFloatData.csv 2 float ZValues.bin")
        print("    text_to_binary.py ExampleTable.txt 5 long Dates.bin '|'")
        return 1
    print(f"Reading column {argv[2]}, of type {argv[3]}...")
    n = convert(argv[1], argv[2], argv[3], argv[4], argv[5] if len(argv) == 6 else ",")
    print(f"Wrote {n} {argv[3]}s to {argv[4]}" if n else "Wrote no data")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
