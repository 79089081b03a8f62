"""Host-side batch plumbing over the C ABI (mirrors the reference's harness objects).

``DeviceBatch`` is the analogue of the reference's ``BatchData``
(benchmarks/benchmark_template_chunked.cuh:162-264, examples/BatchData.h:44-112):
one device slab holding all chunks plus device arrays of chunk pointers and
sizes. ``BatchedCodec`` issues the ``nvcompBatched<Fmt>*`` calls exactly as
``run_benchmark_template`` does (benchmark_template_chunked.cuh:420-451,494-530).

Device memory comes from a small "device" object: ``TorchDevice`` (torch on
ROCm: HBM allocations, HIP stream) here; the CPU-emulation tests supply a
numpy-backed one with the same five methods. No codec logic lives in Python.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from ._lib import OPTS, NvcompStatus


class TorchDevice:
    """HBM buffers and the current HIP stream of one GPU, via torch."""

    def __init__(self, device: str = "cuda:0") -> None:
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("TorchDevice needs a GPU (torch.cuda.is_available() is False)")
        self.torch = torch
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)

    def empty(self, nbytes: int):
        return self.torch.empty(max(int(nbytes), 1), dtype=self.torch.uint8, device=self.device)

    def upload(self, host: np.ndarray):
        host = np.ascontiguousarray(host).view(np.uint8).reshape(-1)
        buf = self.empty(host.size)
        if host.size:
            buf[: host.size].copy_(self.torch.from_numpy(host))
        return buf

    def download(self, buf, nbytes: Optional[int] = None) -> np.ndarray:
        n = buf.numel() if nbytes is None else int(nbytes)
        return buf[:n].cpu().numpy()

    def ptr(self, buf) -> int:
        return int(buf.data_ptr())

    def stream(self) -> int:
        return int(self.torch.cuda.current_stream(self.device).cuda_stream)

    def synchronize(self) -> None:
        self.torch.cuda.synchronize(self.device)


@dataclass
class DeviceBatch:
    """A batch of chunks resident on the device."""

    slab: object          # device buffer holding every chunk
    ptrs: object          # device array of uint64 chunk pointers
    sizes: object         # device array of uint64 chunk sizes (or capacities)
    offsets: np.ndarray   # host copy: chunk start offsets inside the slab
    host_sizes: np.ndarray
    count: int

    @property
    def total_bytes(self) -> int:
        return int(self.host_sizes.sum())


def _layout(sizes: np.ndarray, align: int, stride: Optional[int]) -> Tuple[np.ndarray, int]:
    n = len(sizes)
    offsets = np.zeros(n, dtype=np.int64)
    if stride is not None:
        offsets = np.arange(n, dtype=np.int64) * int(stride)
        return offsets, int(stride) * n
    pos = 0
    for i in range(n):
        pos = (pos + align - 1) // align * align
        offsets[i] = pos
        pos += int(sizes[i])
    return offsets, pos


def make_batch(dev, chunks: Sequence[np.ndarray], align: int = 1, stride: Optional[int] = None,
               base_misalign: int = 0) -> DeviceBatch:
    """Upload chunks into one slab. align=1 packs them tight (examples/BatchData.h:97-103),
    align=8 is the benchmark layout (benchmark_template_chunked.cuh:181-183)."""
    sizes = np.array([c.size for c in chunks], dtype=np.uint64)
    offsets, total = _layout(sizes, align, stride)
    host = np.zeros(total + base_misalign + 16, dtype=np.uint8)
    for c, o in zip(chunks, offsets):
        host[base_misalign + o: base_misalign + o + c.size] = np.asarray(c).view(np.uint8).reshape(-1)
    slab = dev.upload(host)
    base = dev.ptr(slab) + base_misalign
    ptrs = dev.upload((offsets.astype(np.uint64) + np.uint64(base)).view(np.uint8))
    return DeviceBatch(slab, ptrs, dev.upload(sizes.view(np.uint8)), offsets + base_misalign, sizes, len(chunks))


def empty_batch(dev, capacities: Sequence[int], align: int = 1, stride: Optional[int] = None,
                fill: Optional[int] = None, base_misalign: int = 0) -> DeviceBatch:
    """Output batch: one slab with a slot of capacities[i] bytes per chunk."""
    sizes = np.asarray(capacities, dtype=np.uint64)
    offsets, total = _layout(sizes, align, stride)
    if fill is None:
        slab = dev.empty(total + base_misalign + 16)
    else:
        slab = dev.upload(np.full(total + base_misalign + 16, fill, dtype=np.uint8))
    base = dev.ptr(slab) + base_misalign
    ptrs = dev.upload((offsets.astype(np.uint64) + np.uint64(base)).view(np.uint8))
    return DeviceBatch(slab, ptrs, dev.upload(sizes.view(np.uint8)), offsets + base_misalign, sizes, len(sizes))


def read_batch(dev, batch: DeviceBatch, sizes: Optional[Sequence[int]] = None) -> List[np.ndarray]:
    host = dev.download(batch.slab)
    sizes = batch.host_sizes if sizes is None else sizes
    return [host[int(o): int(o) + int(s)].copy() for o, s in zip(batch.offsets, sizes)]


class BatchedCodec:
    """The six ``nvcompBatched<Fmt>*`` entry points of one format."""

    def __init__(self, lib: C.CDLL, dev, fmt: str = "LZ4", opts=None) -> None:
        self.lib, self.dev, self.fmt = lib, dev, fmt
        self._p = "nvcompBatched" + fmt
        if fmt == "Gzip":  # include/nvcomp/gzip.h: decompression only, no options
            self.opts_t = self.opts = None
            return
        self.opts_t = OPTS[fmt]
        if opts is None:
            opts = {"LZ4": (0,), "Snappy": (0,), "Cascaded": (4096, 4, 2, 1, 1), "Bitcomp": (0, 1), "ANS": (0,),
                    "Deflate": (0,)}[fmt]
        self.opts = opts if isinstance(opts, self.opts_t) else self.opts_t(*opts)

    def _fn(self, name: str):
        return getattr(self.lib, self._p + name)

    @staticmethod
    def _check(rc: int, what: str) -> None:
        if rc != NvcompStatus.Success:
            raise RuntimeError(f"{what} returned {rc}")

    # -- size queries (host only) --
    def compress_temp_size(self, batch_size: int, max_chunk: int) -> int:
        out = C.c_size_t(0)
        self._check(self._fn("CompressGetTempSize")(batch_size, max_chunk, self.opts, C.byref(out)),
                    self._p + "CompressGetTempSize")
        return out.value

    def max_compressed_size(self, max_chunk: int) -> int:
        out = C.c_size_t(0)
        self._check(self._fn("CompressGetMaxOutputChunkSize")(max_chunk, self.opts, C.byref(out)),
                    self._p + "CompressGetMaxOutputChunkSize")
        return out.value

    def decompress_temp_size(self, batch_size: int, max_chunk: int) -> int:
        out = C.c_size_t(0)
        self._check(self._fn("DecompressGetTempSize")(batch_size, max_chunk, C.byref(out)),
                    self._p + "DecompressGetTempSize")
        return out.value

    # -- raw async calls on device batches (what bench.py times) --
    def compress_async(self, src: DeviceBatch, dst: DeviceBatch, max_chunk: int, temp, temp_bytes: int) -> int:
        d = self.dev
        return self._fn("CompressAsync")(
            d.ptr(src.ptrs), d.ptr(src.sizes), max_chunk, src.count, d.ptr(temp) if temp is not None else None,
            temp_bytes, d.ptr(dst.ptrs), d.ptr(dst.sizes), self.opts, d.stream())

    def decompress_async(self, comp: DeviceBatch, out: DeviceBatch, actual, statuses, temp, temp_bytes: int) -> int:
        d = self.dev
        return self._fn("DecompressAsync")(
            d.ptr(comp.ptrs), d.ptr(comp.sizes), d.ptr(out.sizes),
            d.ptr(actual) if actual is not None else None, comp.count,
            d.ptr(temp) if temp is not None else None, temp_bytes, d.ptr(out.ptrs),
            d.ptr(statuses) if statuses is not None else None, d.stream())

    def get_decompress_size_async(self, comp: DeviceBatch, sizes_out) -> int:
        d = self.dev
        return self._fn("GetDecompressSizeAsync")(d.ptr(comp.ptrs), d.ptr(comp.sizes), d.ptr(sizes_out), comp.count,
                                                  d.stream())

    # -- convenience round trips on host chunk lists (tests) --
    def compress(self, chunks: Sequence[np.ndarray], in_align: int = 8, max_chunk: Optional[int] = None) -> List[np.ndarray]:
        d = self.dev
        n = len(chunks)
        if max_chunk is None:  # tests may declare less than the largest chunk: such a chunk must come back with size 0
            max_chunk = max([c.size for c in chunks] + [1])
        src = make_batch(d, chunks, align=in_align)
        max_out = self.max_compressed_size(max_chunk)
        dst = empty_batch(d, [max_out] * n, stride=max_out)
        tb = self.compress_temp_size(n, max_chunk)
        temp = d.empty(tb) if tb else None
        self._check(self.compress_async(src, dst, max_chunk, temp, tb), self._p + "CompressAsync")
        d.synchronize()
        sizes = d.download(dst.sizes).view(np.uint64)[:n]
        assert (sizes <= max_out).all(), "compressor overran its declared bound"
        return read_batch(d, dst, sizes)

    def decompress(self, comp_chunks: Sequence[np.ndarray], capacities: Sequence[int], checked: bool = True,
                   want_actual: bool = True, comp_align: int = 1, out_align: int = 1, canary: bool = True,
                   base_misalign: int = 0):
        """Returns (outputs, actual_sizes or None, statuses or None). Output slots are
        followed by canary bytes that must survive (no write past the capacity)."""
        d = self.dev
        n = len(comp_chunks)
        comp = make_batch(d, comp_chunks, align=comp_align, base_misalign=base_misalign)
        pad = 32 if canary else 0
        caps = [int(c) for c in capacities]
        out = empty_batch(d, [c + pad for c in caps], align=out_align, fill=0xA5, base_misalign=base_misalign)
        # the API sees the true capacities, not the padded slots
        out.sizes = d.upload(np.asarray(caps, dtype=np.uint64).view(np.uint8))
        actual = d.upload(np.full(n, 0xDEADBEEF, dtype=np.uint64).view(np.uint8)) if want_actual else None
        statuses = d.upload(np.full(n, -1, dtype=np.int32).view(np.uint8)) if checked else None
        max_chunk = max(caps + [1])
        tb = self.decompress_temp_size(n, max_chunk)
        temp = d.empty(tb) if tb else None
        self._check(self.decompress_async(comp, out, actual, statuses, temp, tb), self._p + "DecompressAsync")
        d.synchronize()
        host = d.download(out.slab)
        outs = []
        for o, c in zip(out.offsets, caps):
            o = int(o)
            outs.append(host[o: o + c].copy())
            if canary:
                assert (host[o + c: o + c + pad] == 0xA5).all(), "decoder wrote past the output capacity"
        act = d.download(actual).view(np.uint64)[:n].copy() if want_actual else None
        st = d.download(statuses).view(np.int32)[:n].copy() if checked else None
        return outs, act, st

    def get_decompress_size(self, comp_chunks: Sequence[np.ndarray], comp_align: int = 1) -> np.ndarray:
        d = self.dev
        n = len(comp_chunks)
        comp = make_batch(d, comp_chunks, align=comp_align)
        sizes = d.upload(np.zeros(n, dtype=np.uint64).view(np.uint8))
        self._check(self.get_decompress_size_async(comp, sizes), self._p + "GetDecompressSizeAsync")
        d.synchronize()
        return d.download(sizes).view(np.uint64)[:n].copy()
