"""The drop-in boundary: libnvcomp.so loads without a GPU and exports every function that
include/nvcomp/*.h declares; option structs and enums have the layout the reference's call
sites pin. No compute calls here."""
import ctypes as C
import os
import re

import pytest

import nvcomp_amd
from nvcomp_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = set()
    for h in ("lz4.h", "snappy.h", "cascaded.h", "bitcomp.h", "ans.h", "deflate.h", "gzip.h"):
        text = open(os.path.join(REPO, "include", "nvcomp", h)).read()
        names |= set(re.findall(r"nvcompStatus_t\s+(nvcompBatched\w+)\s*\(", text))
    return sorted(names)


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(nvcomp_amd.LIB_PATH):
        nvcomp_amd.build_library()
    return C.CDLL(nvcomp_amd.LIB_PATH)


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    # six entry points per format + the *GetTempSizeEx pair (CHANGELOG.md:36-41, 114-117); gzip: decompression only
    assert len(names) == 8 * 6 + 4
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_host_only_queries_work_without_gpu(lib):
    _lib.declare(lib)
    out = C.c_size_t(0)
    assert lib.nvcompBatchedLZ4CompressGetMaxOutputChunkSize(65536, _lib.LZ4Opts(0), C.byref(out)) == 0
    assert out.value == 65809  # == LZ4_compressBound(65536), BASELINE.md section 2
    assert lib.nvcompBatchedSnappyCompressGetMaxOutputChunkSize(65536, _lib.SnappyOpts(0), C.byref(out)) == 0
    assert out.value == 76490  # == snappy::MaxCompressedLength(65536)
    assert lib.nvcompBatchedLZ4DecompressGetTempSize(1000, 65536, C.byref(out)) == 0
    assert lib.nvcompBatchedCascadedDecompressGetTempSize(1000, 65536, C.byref(out)) == 0 and out.value == 4000
    assert lib.nvcompBatchedANSCompressGetMaxOutputChunkSize(65536, _lib.ANSOpts(0), C.byref(out)) == 0
    assert out.value == 65552  # stored form + 12-byte header, rounded to 8
    assert lib.nvcompBatchedBitcompCompressGetMaxOutputChunkSize(65536, _lib.BitcompOpts(0, 5), C.byref(out)) == 0
    assert out.value == 12 + 8 * (32 + 8192) + 4  # 8 full blocks of 2048 uint32 at width 32, rounded to 8
    assert lib.nvcompBatchedLZ4CompressGetTempSize(10, 65536, _lib.LZ4Opts(0), None) == _lib.NvcompStatus.ErrorInvalidValue
    assert lib.nvcompBatchedDeflateDecompressGetTempSize(1000, 65536, C.byref(out)) == 0 and out.value == 0
    assert lib.nvcompBatchedGzipDecompressGetTempSize(1000, 65536, C.byref(out)) == 0 and out.value == 0
    assert lib.nvcompBatchedDeflateCompressGetMaxOutputChunkSize(65536, _lib.DeflateOpts(0), C.byref(out)) == 0
    assert out.value >= 65536 + 10
    assert lib.nvcompBatchedDeflateCompressGetMaxOutputChunkSize(65537, _lib.DeflateOpts(0), C.byref(out)) \
        == _lib.NvcompStatus.ErrorChunkSizeTooLarge  # benchmarks/benchmark_deflate_chunked.cu:53-63
    assert lib.nvcompBatchedDeflateCompressGetTempSize(10, 65536, _lib.DeflateOpts(3), C.byref(out)) \
        == _lib.NvcompStatus.ErrorInvalidValue  # "Deflate algorithm must be 0, 1, or 2" (:43)


def test_struct_and_enum_layout():
    assert C.sizeof(_lib.LZ4Opts) == 4 and C.sizeof(_lib.SnappyOpts) == 4
    assert C.sizeof(_lib.CascadedOpts) == 24 and _lib.CascadedOpts.type.offset == 8
    assert C.sizeof(_lib.BitcompOpts) == 8 and _lib.BitcompOpts.data_type.offset == 4 and C.sizeof(_lib.ANSOpts) == 4
    assert C.sizeof(_lib.DeflateOpts) == 4  # {int algo}: benchmarks/benchmark_deflate_chunked.cu:32,47
    text = open(os.path.join(REPO, "include", "nvcomp", "shared_types.h")).read()
    for name, val in (("nvcompSuccess", 0), ("nvcompErrorCannotDecompress", 12), ("nvcompErrorBadChecksum", 13),
                      ("nvcompErrorAlignment", 17), ("NVCOMP_TYPE_CHAR", 0), ("NVCOMP_TYPE_ULONGLONG", 7)):
        assert re.search(rf"{name}\s*=\s*{val}\b", text), name
    assert re.search(r"NVCOMP_TYPE_BITS\s*=\s*0xff", text)


def test_header_compiles_as_c():
    import subprocess
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write('#include "nvcomp.h"\nint main(void){nvcompBatchedLZ4Opts_t o = nvcompBatchedLZ4DefaultOpts; return (int)o.data_type;}\n')
        subprocess.run(["gcc", "-std=c99", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(REPO, "include"), "-I", "/opt/rocm/include",
                        "-c", src, "-o", os.path.join(d, "t.o")], check=True)


def test_call_logging_env(tmp_path):
    """NVCOMP_LOG_LEVEL / NVCOMP_LOG_FILE (reference README.md:79-88): level 3 logs every low-level call."""
    import subprocess
    import sys

    log = tmp_path / "calls.log"
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')\n"
        "import conftest, numpy as np\n"
        "be = conftest.Backend('emu', conftest.emu_library(), conftest.HostDevice())\n"
        "c = be.codec('LZ4').compress([np.arange(5000, dtype=np.uint8)])\n"
        "be.codec('LZ4').decompress(c, [5000])\n" % (REPO, REPO))
    env = dict(os.environ, NVCOMP_LOG_LEVEL="3", NVCOMP_LOG_FILE=str(log))
    subprocess.run([sys.executable, "-c", code], check=True, env=env)
    text = log.read_text()
    assert "nvcompBatchedLZ4CompressAsync(batch_size=1" in text and "nvcompBatchedLZ4DecompressAsync(batch_size=1" in text


def test_no_runtime_kernel_switches():
    """The shipped library decides nothing from the environment except logging (a drop-in must not have
    a change-my-decoder variable); A/B kernels are -D builds of scripts/build_variants.sh."""
    import glob

    offenders = []
    for path in glob.glob(os.path.join(REPO, "nvcomp_amd", "csrc", "**", "*.h*"), recursive=True):
        if path.endswith(os.path.join("common", "log.h")):
            continue
        if "getenv" in open(path).read():
            offenders.append(os.path.relpath(path, REPO))
    assert not offenders, offenders
    # and the old variable names do nothing: the binary does not even contain them
    blob = open(os.path.join(REPO, "nvcomp_amd", "lib", "libnvcomp.so"), "rb").read() if os.path.exists(
        os.path.join(REPO, "nvcomp_amd", "lib", "libnvcomp.so")) else b""
    assert b"NVCOMP_AMD_LZ4_DECODE" not in blob and b"NVCOMP_AMD_SNAPPY_DECODE" not in blob


def test_no_process_wide_tuning_state():
    """a drop-in library has no setters. Which kernel a batch takes depends on the arguments of the call
    alone (compile-time thresholds, common/lz_launch.hip.h), and the temp-size queries are pure functions of theirs."""
    import re
    import subprocess

    so = os.path.join(REPO, "nvcomp_amd", "lib", "libnvcomp.so")
    if not os.path.exists(so):
        pytest.skip("library not built")
    syms = subprocess.run(["nm", "-D", "--defined-only", so], check=True, capture_output=True, text=True).stdout
    assert not re.findall(r"nvcompAmdSet\w*", syms), "the library must export no nvcompAmdSet* knobs"
    hdr = open(os.path.join(REPO, "include", "nvcomp", "amd_ext.h")).read()
    assert "nvcompAmdSet" not in hdr


def test_cmake_package_exports_nvcomp_target(tmp_path):
    """find_package(nvcomp 3.0.3 REQUIRED) + nvcomp::nvcomp, as the reference's callers write it
    (CMakeLists.txt:18, cmake/nvcomp-config.cmake.in:25-26, benchmarks/CMakeLists.txt:28): a consumer configures, builds
    and runs (host-only entry points)."""
    import shutil
    import subprocess

    if shutil.which("cmake") is None or not os.path.exists(os.path.join(REPO, "nvcomp_amd", "lib", "libnvcomp.so")):
        pytest.skip("cmake or the built library is missing")
    build = tmp_path / "b"
    subprocess.run(["cmake", "-S", os.path.join(REPO, "tests", "cmake_consumer"), "-B", str(build),
                    f"-Dnvcomp_DIR={REPO}/cmake", "-DCMAKE_BUILD_TYPE=Release"], check=True, capture_output=True)
    subprocess.run(["cmake", "--build", str(build)], check=True, capture_output=True)
    out = subprocess.run([str(build / "consumer")], check=True, capture_output=True, text=True).stdout
    assert "LZ4 bound for 64 KiB = 65809" in out


def test_bench_names_the_kernel_the_library_launches():
    """bench.py's `roofline.kernel` (and the PMC traffic records keyed by it) name the kernel by batch size; the
    thresholds live in common/lz_launch.hip.h as compile-time constants -- the two must not drift apart."""
    import re
    import sys

    sys.path.insert(0, REPO)
    import bench

    src = open(os.path.join(REPO, "nvcomp_amd", "csrc", "common", "lz_launch.hip.h")).read()
    team = int(re.search(r"#define NVCOMP_LZ_TEAM_MAX_BATCH (\d+)", src).group(1))
    pair = int(re.search(r"#define NVCOMP_LZ_PAIR_MAX_BATCH (\d+)", src).group(1))
    for algo in ("lz4", "snappy"):
        assert bench.lz_decode_kernel(algo, team) == f"{algo}_decompress_team_kernel"
        assert bench.lz_decode_kernel(algo, team + 1) == f"{algo}_decompress_pair_kernel"
        assert bench.lz_decode_kernel(algo, pair) == f"{algo}_decompress_pair_kernel"
        assert bench.lz_decode_kernel(algo, pair + 1) == f"{algo}_decompress_window_kernel"
