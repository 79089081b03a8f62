"""ctypes binding of libnvcomp.so (the C ABI declared in include/nvcomp/*.h)."""
from __future__ import annotations

import ctypes as C
import enum
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libnvcomp.so")
CSRC = os.path.join(_HERE, "csrc")


class NvcompStatus(enum.IntEnum):
    Success = 0
    ErrorInvalidValue = 10
    ErrorNotSupported = 11
    ErrorCannotDecompress = 12
    ErrorBadChecksum = 13
    ErrorCannotVerifyChecksums = 14
    ErrorOutputBufferTooSmall = 15
    ErrorWrongHeaderLength = 16
    ErrorAlignment = 17
    ErrorChunkSizeTooLarge = 18
    ErrorCudaError = 1000
    ErrorInternal = 10000


class NvcompType(enum.IntEnum):
    CHAR = 0
    UCHAR = 1
    SHORT = 2
    USHORT = 3
    INT = 4
    UINT = 5
    LONGLONG = 6
    ULONGLONG = 7
    BITS = 0xFF


class LZ4Opts(C.Structure):
    _fields_ = [("data_type", C.c_int)]


class SnappyOpts(C.Structure):
    _fields_ = [("reserved", C.c_int)]


class CascadedOpts(C.Structure):
    _fields_ = [("chunk_size", C.c_size_t), ("type", C.c_int), ("num_RLEs", C.c_int), ("num_deltas", C.c_int),
                ("use_bp", C.c_int)]


class BitcompOpts(C.Structure):
    _fields_ = [("algorithm_type", C.c_int), ("data_type", C.c_int)]


class ANSOpts(C.Structure):
    _fields_ = [("type", C.c_int)]


class DeflateOpts(C.Structure):
    _fields_ = [("algo", C.c_int)]


OPTS = {"LZ4": LZ4Opts, "Snappy": SnappyOpts, "Cascaded": CascadedOpts, "Bitcomp": BitcompOpts, "ANS": ANSOpts,
        "Deflate": DeflateOpts}
FORMATS = tuple(OPTS)

# Every symbol include/nvcomp/{lz4,snappy,cascaded,bitcomp,ans,deflate}.h declares, per format.
ENTRY_POINTS = (
    "CompressGetTempSize",
    "CompressGetMaxOutputChunkSize",
    "CompressAsync",
    "DecompressGetTempSize",
    "DecompressAsync",
    "GetDecompressSizeAsync",
)
EXTRA_ENTRY_POINTS = {
    "LZ4": ("CompressGetTempSizeEx", "DecompressGetTempSizeEx"),
    "Snappy": ("CompressGetTempSizeEx", "DecompressGetTempSizeEx"),
    "Cascaded": ("CompressGetTempSizeEx", "DecompressGetTempSizeEx"),
    "Bitcomp": ("CompressGetTempSizeEx", "DecompressGetTempSizeEx"),
    "ANS": ("CompressGetTempSizeEx", "DecompressGetTempSizeEx"),
    "Deflate": ("CompressGetTempSizeEx", "DecompressGetTempSizeEx"),
}
# include/nvcomp/gzip.h: decompression only
GZIP_ENTRY_POINTS = ("DecompressGetTempSize", "DecompressGetTempSizeEx", "DecompressAsync", "GetDecompressSizeAsync")


def build_library(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into nvcomp_amd/lib/libnvcomp.so (hipcc; no GPU needed)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.run(["make", "-C", CSRC, "-j8"], check=True, stdout=out)
    return LIB_PATH


def declare(lib: C.CDLL, formats=FORMATS) -> C.CDLL:
    """Attach argtypes/restype for the formats the library exports."""
    vp, sz, szp = C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)
    for fmt in formats:
        opts = OPTS[fmt]
        pre = "nvcompBatched" + fmt
        if not hasattr(lib, pre + "DecompressAsync"):
            continue
        getattr(lib, pre + "CompressGetTempSize").argtypes = [sz, sz, opts, szp]
        getattr(lib, pre + "CompressGetMaxOutputChunkSize").argtypes = [sz, opts, szp]
        getattr(lib, pre + "CompressAsync").argtypes = [vp, vp, sz, sz, vp, sz, vp, vp, opts, vp]
        getattr(lib, pre + "DecompressGetTempSize").argtypes = [sz, sz, szp]
        getattr(lib, pre + "DecompressAsync").argtypes = [vp, vp, vp, vp, sz, vp, sz, vp, vp, vp]
        getattr(lib, pre + "GetDecompressSizeAsync").argtypes = [vp, vp, vp, sz, vp]
        for name in ENTRY_POINTS + EXTRA_ENTRY_POINTS[fmt]:
            getattr(lib, pre + name).restype = C.c_int
        if "CompressGetTempSizeEx" in EXTRA_ENTRY_POINTS[fmt]:
            getattr(lib, pre + "CompressGetTempSizeEx").argtypes = [sz, sz, opts, szp, sz]
            getattr(lib, pre + "DecompressGetTempSizeEx").argtypes = [sz, sz, szp, sz]
    if hasattr(lib, "nvcompBatchedGzipDecompressAsync"):  # include/nvcomp/gzip.h
        lib.nvcompBatchedGzipDecompressGetTempSize.argtypes = [sz, sz, szp]
        lib.nvcompBatchedGzipDecompressGetTempSizeEx.argtypes = [sz, sz, szp, sz]
        lib.nvcompBatchedGzipDecompressAsync.argtypes = [vp, vp, vp, vp, sz, vp, sz, vp, vp, vp]
        lib.nvcompBatchedGzipGetDecompressSizeAsync.argtypes = [vp, vp, vp, sz, vp]
        for name in GZIP_ENTRY_POINTS:
            getattr(lib, "nvcompBatchedGzip" + name).restype = C.c_int
    if hasattr(lib, "nvcompAmdBatchedPackAsync"):  # include/nvcomp/amd_ext.h
        lib.nvcompAmdBatchedPackAsync.argtypes = [vp, vp, sz, vp, sz, vp, vp]
    for fmt in ("LZ4", "Snappy"):
        fn = getattr(lib, f"nvcompAmdBatched{fmt}TokenIndexAsync", None)
        if fn is not None:
            fn.argtypes, fn.restype = [vp, vp, sz, vp, vp, vp], C.c_int
    return lib


def load_library(path: str = LIB_PATH) -> C.CDLL:
    """Load the HIP library. Raises if it has not been built: there is no fallback.
    NVCOMP_AMD_LIB selects an alternative build of the same HIP library (A/B tuning builds)."""
    if path == LIB_PATH:
        path = os.environ.get("NVCOMP_AMD_LIB", path)
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). nvcomp_amd has no CPU fallback.")
    # The torch wheel bundles its own HIP runtime (torch/lib/libamdhip64.so). Device
    # pointers and streams handed to libnvcomp.so come from torch, so torch's runtime
    # must be the one already resident when libnvcomp.so resolves its libamdhip64
    # dependency; loading in the other order puts two runtimes in one process and
    # every launch on a torch stream fails (seen on MI355X as hipError -> status 1000).
    import torch  # noqa: F401

    return declare(C.CDLL(path))
