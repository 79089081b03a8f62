"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product (``nvcomp_amd``) must never do so.

Two libraries:
  * ``oracle/liboracle.so``         the C restatement ("port") of the codecs
  * ``oracle/_ref/libcpucodecs.so`` shim over the container's liblz4 / snappy,
    the third-party codecs the reference pins its wire formats to
    (examples/lz4_cpu_compression.cu:61-66, examples/lz4_cpu_decompression.cu:143-147).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PORT = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libcpucodecs.so")

LZ4_DEC, SNAPPY_DEC, LZ4_ENC, SNAPPY_ENC, LZ4_ENC_HC = 0, 1, 2, 3, 4
ZLIB_INFLATE, ZLIB_DEFLATE_1, ZLIB_DEFLATE_9 = 5, 6, 7  # the _ref shim only (zlib: the CPU peer of the DEFLATE path)
LIBDEFLATE_DEC, LIBDEFLATE_ENC_6 = 8, 9  # the _ref shim only (libdeflate: the reference's algo 0 of the same examples)
CASCADED_DEC, BITCOMP_DEC, ANS_DEC = 4, 5, 6  # oracle_batch_run only (the port library; 4 means HC in the reference shim)

_u8p = C.POINTER(C.c_uint8)
_szp = C.POINTER(C.c_size_t)


def build(force: bool = False) -> None:
    """Compile the oracle with gcc (seconds). The _ref shim is built only where
    the container's liblz4/snappy development files exist."""
    stale = os.path.exists(_REF) and os.path.getmtime(_REF) < os.path.getmtime(os.path.join(_HERE, "ref_shim.c"))
    if force or stale or not os.path.exists(_PORT) or (not os.path.exists(_REF) and os.path.exists("/opt/conda/include/lz4.h")):
        subprocess.run(["make", "-C", _HERE, "all"], check=True, stdout=subprocess.DEVNULL)


def _load(path: str) -> Optional[C.CDLL]:
    try:
        return C.CDLL(path)
    except OSError:
        return None


class _Lib:
    def __init__(self) -> None:
        build()
        self.port = C.CDLL(_PORT)
        self.ref = _load(_REF)
        p = self.port
        for name in ("oracle_lz4_decompress", "oracle_snappy_decompress"):
            f = getattr(p, name)
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, _szp]
        p.oracle_lz4_decompressed_size.restype = C.c_size_t
        p.oracle_lz4_decompressed_size.argtypes = [C.c_void_p, C.c_size_t]
        p.oracle_snappy_decompressed_size.restype = C.c_int
        p.oracle_snappy_decompressed_size.argtypes = [C.c_void_p, C.c_size_t, _szp]
        for name in ("oracle_lz4_compress", "oracle_snappy_compress"):
            f = getattr(p, name)
            f.restype = C.c_size_t
            f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        for name in ("oracle_lz4_compress_bound", "oracle_snappy_compress_bound"):
            f = getattr(p, name)
            f.restype = C.c_size_t
            f.argtypes = [C.c_size_t]
        p.oracle_batch_run.restype = C.c_double
        p.oracle_batch_run.argtypes = [
            C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
            C.POINTER(C.c_int)]
        if hasattr(p, "oracle_cascaded_compress"):
            p.oracle_cascaded_compress.restype = C.c_size_t
            p.oracle_cascaded_compress.argtypes = [
                C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int]
            p.oracle_cascaded_decompress.restype = C.c_int
            p.oracle_cascaded_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, _szp]
            p.oracle_cascaded_max_compressed.restype = C.c_size_t
            p.oracle_cascaded_max_compressed.argtypes = [C.c_size_t, C.c_size_t, C.c_int]
        if hasattr(p, "oracle_bitcomp_compress"):
            p.oracle_bitcomp_compress.restype = C.c_size_t
            p.oracle_bitcomp_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
            p.oracle_bitcomp_decompress.restype = C.c_int
            p.oracle_bitcomp_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, _szp]
            p.oracle_bitcomp_max_compressed.restype = C.c_size_t
            p.oracle_bitcomp_max_compressed.argtypes = [C.c_size_t, C.c_int]
        if hasattr(p, "oracle_ans_compress"):
            p.oracle_ans_compress.restype = C.c_size_t
            p.oracle_ans_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
            p.oracle_ans_decompress.restype = C.c_int
            p.oracle_ans_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, _szp]
            p.oracle_ans_max_compressed.restype = C.c_size_t
            p.oracle_ans_max_compressed.argtypes = [C.c_size_t]
        r = self.ref
        if r is not None:
            for name in ("ref_lz4_decompress", "ref_lz4_compress", "ref_snappy_decompress", "ref_snappy_compress"):
                f = getattr(r, name)
                f.restype = C.c_int
                f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, _szp]
            r.ref_lz4_compress_hc.restype = C.c_int
            r.ref_lz4_compress_hc.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, _szp]
            r.ref_snappy_uncompressed_length.restype = C.c_int
            r.ref_snappy_uncompressed_length.argtypes = [C.c_void_p, C.c_size_t, _szp]
            r.ref_lz4_bound.restype = C.c_size_t
            r.ref_lz4_bound.argtypes = [C.c_size_t]
            r.ref_snappy_bound.restype = C.c_size_t
            r.ref_snappy_bound.argtypes = [C.c_size_t]
            r.ref_lz4_version.restype = C.c_int
            if hasattr(r, "ref_libdeflate_decompress"):
                for name in ("ref_libdeflate_decompress", "ref_libdeflate_compress"):
                    f = getattr(r, name)
                    f.restype = C.c_int
                    f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, _szp]
            r.ref_batch_run.restype = C.c_double
            r.ref_batch_run.argtypes = p.oracle_batch_run.argtypes


_lib: Optional[_Lib] = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib


def have_ref() -> bool:
    return lib().ref is not None


def _as_u8(buf) -> np.ndarray:
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    return np.ascontiguousarray(a.view(np.uint8).reshape(-1))


def _ptr(a: np.ndarray) -> int:
    return a.ctypes.data if a.size else 0


# ---------------------------------------------------------------- single chunk

def _dec(fn, comp, cap: int) -> Tuple[int, np.ndarray]:
    src = _as_u8(comp)
    dst = np.empty(max(cap, 1), dtype=np.uint8)
    out = C.c_size_t(0)
    rc = fn(_ptr(src), src.size, _ptr(dst), cap, C.byref(out))
    return rc, dst[: out.value].copy()


def lz4_decompress(comp, cap: int) -> Tuple[int, np.ndarray]:
    """Port. Returns (rc, bytes); rc 0 = ok."""
    return _dec(lib().port.oracle_lz4_decompress, comp, cap)


def snappy_decompress(comp, cap: int) -> Tuple[int, np.ndarray]:
    return _dec(lib().port.oracle_snappy_decompress, comp, cap)


def lz4_decompressed_size(comp) -> int:
    src = _as_u8(comp)
    return int(lib().port.oracle_lz4_decompressed_size(_ptr(src), src.size))


def snappy_decompressed_size(comp) -> Tuple[int, int]:
    src = _as_u8(comp)
    out = C.c_size_t(0)
    rc = lib().port.oracle_snappy_decompressed_size(_ptr(src), src.size, C.byref(out))
    return rc, out.value


def _enc(fn, bound_fn, raw) -> np.ndarray:
    src = _as_u8(raw)
    cap = int(bound_fn(src.size))
    dst = np.empty(max(cap, 1), dtype=np.uint8)
    n = fn(_ptr(src), src.size, _ptr(dst), cap)
    return dst[:n].copy()


def lz4_compress(raw) -> np.ndarray:
    p = lib().port
    return _enc(p.oracle_lz4_compress, p.oracle_lz4_compress_bound, raw)


def snappy_compress(raw) -> np.ndarray:
    p = lib().port
    return _enc(p.oracle_snappy_compress, p.oracle_snappy_compress_bound, raw)


def lz4_bound(n: int) -> int:
    return int(lib().port.oracle_lz4_compress_bound(n))


def snappy_bound(n: int) -> int:
    return int(lib().port.oracle_snappy_compress_bound(n))


def cascaded_compress(raw, sub_chunk: int = 4096, type_: int = 4, num_rles: int = 2, num_deltas: int = 1,
                      use_bp: int = 1) -> np.ndarray:
    p = lib().port
    src = _as_u8(raw)
    cap = int(p.oracle_cascaded_max_compressed(src.size, sub_chunk, type_))
    dst = np.zeros(max(cap, 1), dtype=np.uint8)
    n = p.oracle_cascaded_compress(_ptr(src), src.size, _ptr(dst), cap, sub_chunk, type_, num_rles, num_deltas, use_bp)
    assert n > 0, "oracle_cascaded_compress rejected its arguments"
    return dst[:n].copy()


def cascaded_decompress(comp, cap: int) -> Tuple[int, np.ndarray]:
    return _dec(lib().port.oracle_cascaded_decompress, comp, cap)


def cascaded_bound(n: int, sub_chunk: int = 4096, type_: int = 4) -> int:
    return int(lib().port.oracle_cascaded_max_compressed(n, sub_chunk, type_))


def bitcomp_compress(raw, algo: int = 0, elem_size: int = 1) -> np.ndarray:
    p = lib().port
    src = _as_u8(raw)
    cap = int(p.oracle_bitcomp_max_compressed(src.size, elem_size))
    dst = np.zeros(max(cap, 1), dtype=np.uint8)
    n = p.oracle_bitcomp_compress(_ptr(src), src.size, _ptr(dst), cap, algo, elem_size)
    assert n > 0, "oracle_bitcomp_compress rejected its arguments"
    return dst[:n].copy()


def bitcomp_decompress(comp, cap: int) -> Tuple[int, np.ndarray]:
    return _dec(lib().port.oracle_bitcomp_decompress, comp, cap)


def ans_compress(raw) -> np.ndarray:
    p = lib().port
    src = _as_u8(raw)
    cap = int(p.oracle_ans_max_compressed(src.size))
    dst = np.zeros(max(cap, 1), dtype=np.uint8)
    n = p.oracle_ans_compress(_ptr(src), src.size, _ptr(dst), cap)
    assert n > 0, "oracle_ans_compress rejected its arguments"
    return dst[:n].copy()


def ans_decompress(comp, cap: int) -> Tuple[int, np.ndarray]:
    return _dec(lib().port.oracle_ans_decompress, comp, cap)


# ------------------------------------------------- liblz4 / libsnappy ("reference")

def ref_lz4_decompress(comp, cap: int) -> Tuple[int, np.ndarray]:
    return _dec(lib().ref.ref_lz4_decompress, comp, cap)


def ref_snappy_decompress(comp, cap: int) -> Tuple[int, np.ndarray]:
    return _dec(lib().ref.ref_snappy_decompress, comp, cap)


def ref_lz4_compress(raw, hc_level: int = 0) -> np.ndarray:
    r = lib().ref
    src = _as_u8(raw)
    cap = int(r.ref_lz4_bound(src.size))
    dst = np.empty(max(cap, 1), dtype=np.uint8)
    out = C.c_size_t(0)
    if hc_level > 0:
        rc = r.ref_lz4_compress_hc(_ptr(src), src.size, _ptr(dst), cap, hc_level, C.byref(out))
    else:
        rc = r.ref_lz4_compress(_ptr(src), src.size, _ptr(dst), cap, C.byref(out))
    assert rc == 0
    return dst[: out.value].copy()


def ref_snappy_compress(raw) -> np.ndarray:
    r = lib().ref
    src = _as_u8(raw)
    cap = int(r.ref_snappy_bound(src.size))
    dst = np.empty(max(cap, 1), dtype=np.uint8)
    out = C.c_size_t(0)
    rc = r.ref_snappy_compress(_ptr(src), src.size, _ptr(dst), cap, C.byref(out))
    assert rc == 0
    return dst[: out.value].copy()


def have_libdeflate() -> bool:
    return have_ref() and hasattr(lib().ref, "ref_libdeflate_decompress")


def ref_libdeflate_compress(buf) -> np.ndarray:
    """libdeflate_deflate_compress at level 6: what examples/deflate_cpu_compression.cu:60-67 (algo 0) writes."""
    src = _as_u8(buf)
    cap = src.size + src.size // 8 + 64
    dst = np.empty(cap, dtype=np.uint8)
    out = C.c_size_t(0)
    rc = lib().ref.ref_libdeflate_compress(_ptr(src), src.size, _ptr(dst), cap, C.byref(out))
    assert rc == 0
    return dst[: out.value].copy()


def ref_libdeflate_decompress(buf, cap: int):
    src = _as_u8(buf)
    dst = np.empty(max(cap, 1), dtype=np.uint8)
    out = C.c_size_t(0)
    rc = lib().ref.ref_libdeflate_decompress(_ptr(src), src.size, _ptr(dst), cap, C.byref(out))
    return rc, dst[: out.value].copy()


# ------------------------------------------------------------------- batches

def batch_run(codec: int, chunks: Sequence[np.ndarray], out_caps: Sequence[int], threads: int = 1,
              repeats: int = 1, use_ref: bool = False) -> Tuple[float, List[np.ndarray], int]:
    """Run one codec over a list of chunks. Returns (best wall seconds, outputs, error count)."""
    l = lib()
    n = len(chunks)
    ins = [_as_u8(c) for c in chunks]
    in_ptrs = (C.c_void_p * n)(*[_ptr(a) for a in ins])
    in_sizes = (C.c_size_t * n)(*[a.size for a in ins])
    offs = np.zeros(n + 1, dtype=np.int64)
    offs[1:] = np.cumsum(np.asarray(out_caps, dtype=np.int64))
    slab = np.empty(max(int(offs[-1]), 1), dtype=np.uint8)
    base = slab.ctypes.data
    out_ptrs = (C.c_void_p * n)(*[base + int(o) for o in offs[:-1]])
    caps = (C.c_size_t * n)(*[int(c) for c in out_caps])
    out_sizes = (C.c_size_t * n)()
    errs = C.c_int(0)
    fn = l.ref.ref_batch_run if use_ref else l.port.oracle_batch_run
    secs = fn(codec, threads, repeats, n, in_ptrs, in_sizes, out_ptrs, caps, out_sizes, C.byref(errs))
    outs = [slab[int(offs[i]): int(offs[i]) + int(out_sizes[i])] for i in range(n)]
    return float(secs), outs, int(errs.value)
