"""The C++ example and harness programs (examples/, benchmarks/) run end to end:
they verify their own results (non-zero exit code on any mismatch), and the harness's
stdout must stay awk-compatible with the reference's scripts (benchmarks/benchmark.sh:28-31)."""
import os
import re
import subprocess

import numpy as np
import pytest

from nvcomp_amd import datasets

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, **kw):
    r = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=600, **kw)
    assert r.returncode == 0, f"{' '.join(cmd)}\n{r.stdout}\n{r.stderr}"
    return r.stdout


def test_hlif_example_on_emulator(tmp_path):
    """high_level_quickstart_example against the host emulation of the kernels (CPU-only)."""
    import conftest

    conftest.emu_library()
    exe = tmp_path / "hlq_emu"
    run(["g++", "-O1", "-std=c++17", "-Itests/emu", "-Iinclude", "-Iexamples", "examples/high_level_quickstart_example.cpp",
         "-o", str(exe), "-Ltests/emu", "-lnvcomp_emu", f"-Wl,-rpath,{REPO}/tests/emu"])
    assert "all scenarios passed" in run([str(exe)])


def test_low_level_example_on_emulator(tmp_path):
    import conftest

    conftest.emu_library()
    exe = tmp_path / "llq_emu"
    run(["g++", "-O1", "-std=c++17", "-Itests/emu", "-Iinclude", "-Iexamples", "examples/low_level_quickstart_example.cpp",
         "-o", str(exe), "-Ltests/emu", "-lnvcomp_emu", f"-Wl,-rpath,{REPO}/tests/emu"])
    assert "round-tripped in place" in run([str(exe)])


def test_crc_is_the_standard_crc32_on_emulator(tmp_path):
    """The managers' per-chunk checksums equal zlib.crc32 (reference examples/standard_crc_checksum.cpp:94-107 compares
    with boost::crc_32_type): the example reads crc_uncomp[] / crc_comp[] out of the container and recomputes both."""
    import conftest

    conftest.emu_library()
    exe = tmp_path / "crc_emu"
    run(["g++", "-O1", "-std=c++17", "-Itests/emu", "-Iinclude", "-Iexamples", "examples/standard_crc_checksum.cpp",
         "-o", str(exe), "-Ltests/emu", "-lnvcomp_emu", "-lz", f"-Wl,-rpath,{REPO}/tests/emu"])
    assert "equals zlib's crc32()" in run([str(exe)])


@pytest.fixture(scope="module")
def built_programs():
    run(["make", "-C", "benchmarks", "-j8"])
    run(["make", "-C", "examples", "-j4"])


@pytest.fixture(scope="module")
def sample_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("data")
    files = {}
    for name, gen in (("table.txt", datasets.table_rows), ("floats.csv", datasets.float_csv), ("col.int32", datasets.int32_column)):
        p = d / name
        gen(300000, 1).tofile(p)
        files[name] = str(p)
    return files


@pytest.mark.gpu
def test_examples_on_gpu(built_programs, sample_files):
    assert "all scenarios passed" in run(["examples/bin/high_level_quickstart_example"])
    assert "round-tripped in place" in run(["examples/bin/low_level_quickstart_example"])
    assert "equals zlib's crc32()" in run(["examples/bin/standard_crc_checksum"])
    out = run(["examples/bin/nvcomp_storage", os.path.join(os.path.dirname(sample_files["table.txt"]), "storage.bin"), "30000000"])
    assert "PASSED: Uncompressed data is identical to the input" in out  # examples/nvcomp_gds.cu's round trip through a file
    for algo in ("0", "1", "2", "gzip"):  # libdeflate (0) / zlib on the CPU -> the DEFLATE / gzip decoder (examples/deflate_cpu_compression.cu,
        out = run(["examples/bin/deflate_cpu_compression", "-a", algo, "-f", sample_files["table.txt"],  # gzip_gpu_decompression.cu)
                   sample_files["floats.csv"]])
        assert "decompression validated" in out
    if os.path.exists(os.path.join(REPO, "examples/bin/lz4_cpu_compression")):
        out = run(["examples/bin/lz4_cpu_compression", "-f", sample_files["table.txt"], sample_files["floats.csv"]])
        assert "decompression validated" in out
        out = run(["examples/bin/lz4_cpu_decompression", "-f", sample_files["table.txt"], sample_files["floats.csv"]])
        assert "decompression validated" in out


@pytest.mark.gpu
@pytest.mark.parametrize("prog,extra", [("benchmark_lz4_chunked", []), ("benchmark_snappy_chunked", []),
                                         ("benchmark_cascaded_chunked", ["-t", "int"]),
                                         ("benchmark_ans_chunked", []), ("benchmark_bitcomp_chunked", ["-t", "int"]),
                                         ("benchmark_deflate_chunked", ["-a", "1"]),
                                         ("benchmark_bitcomp_chunked", ["-t", "int", "-a", "1"])])
def test_chunked_harness_on_gpu(built_programs, sample_files, prog, extra):
    f = sample_files["col.int32"] if "cascaded" in prog or "bitcomp" in prog else sample_files["table.txt"]
    out = run([f"benchmarks/bin/{prog}", "-f", f, "-i", "2", "-x", "4"] + extra)
    lines = out.strip().splitlines()
    assert lines[0] == "----------" and lines[1] == "files: 1"
    assert re.match(r"uncompressed \(B\): \d+$", lines[2])
    assert re.match(r"comp_size: \d+, compressed ratio: \d+\.\d{4}$", lines[3])
    assert re.match(r"compression throughput \(GB/s\): \d+\.\d{4}$", lines[4])
    assert re.match(r"decompression throughput \(GB/s\): \d+\.\d{4}$", lines[5])
    csv = run([f"benchmarks/bin/{prog}", "-f", f, "-c", "true"] + extra)
    assert csv.splitlines()[0].startswith("Files, Duplicate data, Size in MB, Pages")


@pytest.mark.gpu
def test_hlif_and_synth_benchmarks_on_gpu(built_programs, sample_files):
    for fmt in ("lz4", "snappy", "cascaded", "bitcomp", "ans", "deflate"):
        numeric = fmt in ("cascaded", "bitcomp")
        f = sample_files["col.int32"] if numeric else sample_files["table.txt"]
        out = run(["benchmarks/bin/benchmark_hlif", fmt, "-f", f, "-n", "2"] + (["-t", "int"] if numeric else []))
        assert "decompression throughput (GB/s):" in out and "compressed ratio:" in out
    out = run(["benchmarks/bin/benchmark_snappy_synth", "-b", "500", "-w", "2", "-i", "3"])
    assert "decompression throughput (GB/s):" in out
    out = run(["benchmarks/bin/benchmark_lz4_synth", "-b", "4"])
    assert out.count("zeros") == 5 and out.count("random") == 5


@pytest.mark.gpu
def test_allgather_program_oversubscribed(built_programs, sample_files):
    """benchmark_allgather logic on however many GPUs the box has (logical GPUs share devices)."""
    for comp in ("none", "lz4"):
        out = run(["benchmarks/bin/benchmark_allgather", "-f", sample_files["table.txt"], "-g", "2", "-h", "4", "-c", comp,
                   "--oversubscribe"])
        assert float(out.split()[-1]) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("algo", ["lz4", "snappy", "cascaded", "bitcomp", "ans", "deflate"])
def test_bench_contract_on_gpu(algo):
    """bench.py prints exactly one JSON line with the driver's fields plus `roofline` and `cpu_baseline`, measures
    through HIP events on its stream and verifies every decoded byte (small workload: 64 MiB per step)."""
    import json
    import sys

    r = subprocess.run([sys.executable, "bench.py", "--algo", algo, "--steps", "3", "--warmup", "1", "--mib-per-gpu", "64",
                        "--unique-mib", "8"], cwd=REPO, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in res, key
    assert res["n_gpus"] == 1 and res["steps"] == 3 and res["warmup"] == 1 and res["unit"] == "GB/s"
    assert res["value"] > 1.0 and res["config"]["verified"] is True and "workload" in res["config"]
    roof = res["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and 0 < roof["frac"] < 1
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    base = res["cpu_baseline"]
    assert base["value"] > 0 and base["cores"] >= 1 and base["kind"] in ("reference", "port") and base["sample"]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--allgather"], ["--allgather", "--mib-per-gpu", "4096", "--unique-mib", "64"]],
                         ids=["sharded", "allgather", "allgather_4gib"])
def test_bench_under_launcher_with_rccl(extra):
    """The driver's N>1 launch line, with one rank (the box has one GPU) and RCCL forced up: process-group init on
    the device, barrier, max-over-ranks all-reduce, digest gather and (--allgather) the RCCL all-gathers all run
    on the real backend instead of only on gloo (tests/test_multiprocess.py)."""
    import json
    import os
    import sys

    env = dict(os.environ, NVCOMP_AMD_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29641", "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--mib-per-gpu", "64", "--unique-mib", "8", "--no-cpu-baseline", "--no-extras"] + extra  # (later flags win:
    # "allgather_4gib" runs the default 4 GiB-per-GPU buffer plan of the all-gather step once on hardware)
    r = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-8000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 1 and res["steps"] == 2
    if not extra:
        assert res["value"] > 1.0 and len(res["config"]["shard_digests"]) == 1
