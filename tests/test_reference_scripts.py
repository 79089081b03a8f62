"""The reference's OWN benchmark scripts, unmodified, drive this library's harness programs.

benchmarks/benchmark.sh:24-34 and benchmarks/benchmark_all_algorithms.sh:59-199 of the reference run
`./bin/benchmark_<algo>_chunked -f FILE [-t TYPE] [-a VARIANT] [-r R -d D -b B]` from the benchmarks directory and cut four
numbers out of every program's stdout with awk (`^uncompressed `, `compressed ratio:`, `^compression throughput `,
`^decompression throughput `). A user switching libraries keeps those scripts; here they are executed as they are
(read from /root/reference at run time -- nothing of them is copied into this repository) against the harness programs of
benchmarks/*.cpp.

The scripts only exist in the development container, so the test runs here, on the CPU, with the programs linked
against the kernels' host emulation (tests/emu): same sources, same stdout. On a GPU box (no /root/reference) it skips;
the programs' stdout contract itself is checked there by tests/test_programs.py::test_chunked_harness_on_gpu.
"""
import csv
import os
import subprocess

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/benchmarks"
ALGOS = ["lz4", "snappy", "cascaded", "bitcomp", "deflate", "ans"]

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "benchmark.sh")),
                                reason="the reference tree is not on this machine")


@pytest.fixture(scope="module")
def emu_bench_dir(tmp_path_factory):
    """A directory laid out like the reference's benchmarks/: ./bin/benchmark_<algo>_chunked, built for the emulator."""
    import conftest

    conftest.emu_library()
    root = tmp_path_factory.mktemp("bench")
    os.makedirs(root / "bin")
    jobs = []
    for algo in ALGOS:
        cmd = ["g++", "-O1", "-std=c++17", "-Itests/emu", "-Iinclude", "-Ibenchmarks", "-Iexamples",
               f"benchmarks/benchmark_{algo}_chunked.cpp", "-o", str(root / "bin" / f"benchmark_{algo}_chunked"),
               "-Ltests/emu", "-lnvcomp_emu", f"-Wl,-rpath,{REPO}/tests/emu"]
        jobs.append(subprocess.Popen(cmd, cwd=REPO, stderr=subprocess.PIPE))
    for j in jobs:
        _, err = j.communicate()
        assert j.returncode == 0, err.decode()[-2000:]
    return root


def _datasets(dirpath, names):
    from nvcomp_amd import datasets

    gens = [datasets.int32_column, datasets.silesia_style, datasets.float_columns, datasets.table_rows]
    for i, name in enumerate(names):
        data = gens[i % len(gens)](96 * 1024 + 40 * i, i)
        np.asarray(data).tofile(os.path.join(dirpath, name))


def test_benchmark_sh_runs_unchanged(emu_bench_dir, tmp_path):
    """benchmark.sh <algo> <directory>: a header line, then one CSV row per file -- name, bytes, ratio, two throughputs."""
    data = tmp_path / "data"
    os.makedirs(data)
    _datasets(str(data), ["a_column.bin", "b_mix.bin"])
    for algo in ("lz4", "snappy", "ans"):
        r = subprocess.run(["bash", os.path.join(REF, "benchmark.sh"), algo, str(data)], cwd=str(emu_bench_dir),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        rows = list(csv.reader(r.stdout.strip().splitlines(), skipinitialspace=True))
        assert rows[0] == ["dataset", "uncompressed bytes", "compression ratio", "compression throughput (GB/s)",
                           "decompression throughput (GB/s)"]
        assert [row[0] for row in rows[1:]] == ["a_column.bin", "b_mix.bin"]
        for row in rows[1:]:
            assert len(row) == 5, row
            assert int(row[1]) == os.path.getsize(data / row[0])
            assert float(row[2]) > 0.9 and float(row[3]) >= 0 and float(row[4]) >= 0, row
        assert float(rows[1][2]) > 1.5, "the int32 column compresses with every one of these codecs"


def test_benchmark_all_algorithms_sh_runs_unchanged(emu_bench_dir, tmp_path):
    """benchmark_all_algorithms.sh <directory> <output.csv> <gpu name>: the four datasets it names, every algorithm it
    lists with the per-dataset flags it chooses (-t int for LZ4 on the integer column, -a 0 / 1 and -t for Bitcomp,
    -r -d -b -t for Cascaded). GDeflate and zstd, which SURVEY.md puts out of scope, have no program: the script's own
    error handling leaves their rows empty and goes on."""
    data = tmp_path / "data"
    os.makedirs(data)
    names = ["mortgage-2009Q2-col0-long.bin", "silesia.tar", "texturecache.tar", "geometrycache.tar"]
    _datasets(str(data), names)
    out = tmp_path / "all.csv"
    r = subprocess.run(["bash", os.path.join(REF, "benchmark_all_algorithms.sh"), str(data), str(out), "MI355X (emulated)"],
                       cwd=str(emu_bench_dir), capture_output=True, text=True, timeout=1700)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = out.read_text().splitlines()
    assert lines[0] == ",,MI355X (emulated)" and lines[1] == ",,Compression Ratio,Compression Throughput,Decompression Throughput"
    text = "\n".join(lines)
    for title in ("Data Analytics: INT Columns", "Silesia", "Graphics: Textures Data", "Graphics: Geometry Data"):
        assert title in text
    rows = [l.split(",") for l in lines[2:] if l.startswith(",")]
    by_algo = {}
    for row in rows:
        by_algo.setdefault(row[1], []).append(row)
    for algo in ("lz4", "snappy", "cascaded", "bitcomp-default", "bitcomp-sparse", "deflate", "ans"):
        assert len(by_algo.get(algo, [])) == 4, (algo, by_algo.keys())
        for row in by_algo[algo]:
            # (the throughputs are those of the host EMULATION, printed with four decimals: a workgroup-per-chunk decode of
            # 512 - 1 024 coroutines can round to 0.0000 GB/s -- the columns must parse, not impress)
            assert float(row[2]) > 0.5 and float(row[3]) >= 0 and float(row[4]) >= 0, row
    # the commands the script composed are echoed: the typed LZ4 run and the cascaded scheme reached our programs
    assert "benchmark_lz4_chunked -f" in r.stdout and "-t int" in r.stdout
    assert "benchmark_cascaded_chunked -f" in r.stdout and "-r 1 -d 0 -b 1 -t longlong" in r.stdout
