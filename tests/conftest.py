"""Test scaffolding.

Two backends run the SAME tests through the SAME C ABI:
  * ``gpu``  -- nvcomp_amd/lib/libnvcomp.so on a real MI355X (marked ``gpu``);
  * ``emu``  -- tests/emu/libnvcomp_emu.so: the library's kernel sources compiled
    for the host against tests/emu (lanes as coroutines). CPU-only debugging aid,
    test infrastructure; it is not a fallback of the product.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


class HostDevice:
    """numpy-backed stand-in for TorchDevice used with the emulation library."""

    def empty(self, nbytes):
        return np.zeros(max(int(nbytes), 1), dtype=np.uint8)

    def upload(self, host):
        return np.ascontiguousarray(host).view(np.uint8).reshape(-1).copy()

    def download(self, buf, nbytes=None):
        return buf[: buf.size if nbytes is None else int(nbytes)].copy()

    def ptr(self, buf):
        return int(buf.data_ptr()) if hasattr(buf, "data_ptr") else buf.ctypes.data

    def stream(self):
        return None

    def synchronize(self):
        pass


_emu_lib = None


def _make_emu(*targets):
    """Build the emulation library; serialised across pytest-xdist workers (they all start by asking for it)."""
    import fcntl

    emu_dir = os.path.join(REPO, "tests", "emu")
    with open(os.path.join(emu_dir, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        subprocess.run(["make", "-C", emu_dir, "-j8", *targets], check=True, stdout=subprocess.DEVNULL)
    return emu_dir


def emu_library():
    global _emu_lib
    if _emu_lib is None:
        from nvcomp_amd import _lib

        emu_dir = _make_emu()
        _emu_lib = _lib.declare(C.CDLL(os.path.join(emu_dir, "libnvcomp_emu.so")))
    return _emu_lib


_emu_path_libs = {}


def emu_path_library(path):
    """The emulation build with one LZ decode path forced whatever the batch size ("chase": one persistent wave per
    chunk, "pair": two waves per chunk, "team": a workgroup per chunk). The product picks by batch size (common/lz_launch.hip.h); these are A/B builds
    of the same sources with another compile-time threshold."""
    if path not in _emu_path_libs:
        from nvcomp_amd import _lib

        emu_dir = _make_emu(path)
        _emu_path_libs[path] = _lib.declare(C.CDLL(os.path.join(emu_dir, f"libnvcomp_emu_{path}.so")))
    return _emu_path_libs[path]


_gpu_path_libs = {}


def gpu_path_library(path):
    if path not in _gpu_path_libs:
        import nvcomp_amd

        _gpu_path_libs[path] = nvcomp_amd.load_library(os.path.join(REPO, "nvcomp_amd", "lib", "alt", f"libnvcomp_{path}.so"))
    return _gpu_path_libs[path]


class Backend:
    def __init__(self, name, lib, dev):
        self.name, self.lib, self.dev = name, lib, dev

    def codec(self, fmt="LZ4", opts=None):
        from nvcomp_amd.batched import BatchedCodec

        return BatchedCodec(self.lib, self.dev, fmt, opts)


@pytest.fixture(scope="session")
def emu():
    return Backend("emu", emu_library(), HostDevice())


@pytest.fixture(scope="session")
def gpu():
    import nvcomp_amd

    return Backend("gpu", nvcomp_amd.load_library(), nvcomp_amd.TorchDevice("cuda:0"))


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def backend(request):
    """The library exactly as shipped: the LZ decoders pick their path by batch size (two waves per chunk for small
    batches, persistent waves above). Tests of the LZ decoders additionally ask for `lz_path`, which swaps in the A/B
    builds that force either path whatever the batch size."""
    b = request.getfixturevalue(request.param)
    return Backend(b.name, b.lib, b.dev)


@pytest.fixture(params=["default", "chase", "pair", "team"])
def lz_path(request, backend):
    """All decode paths of nvcompBatched{LZ4,Snappy}DecompressAsync, whatever the batch size."""
    if request.param == "team" and backend.name == "emu":
        # The shipped thresholds already send every batch of up to 512 chunks -- all an emulated test can afford -- to the
        # workgroup-per-chunk kernels, so `default` IS this path here (same sources, same kernels); a 512- or 1 024-lane
        # workgroup is 8-16 x the coroutines of the other paths and the CPU tier must stay a matter of minutes. The forced
        # build runs on the GPU (where it differs: batches above 512), and tests/test_lz4_decode.py::
        # test_persistent_workgroups covers the eight-wave persistent launch on the emulator.
        pytest.skip("emulator: lz_path = default already runs the workgroup-per-chunk kernels for batches this small")
    if request.param != "default":
        backend.lib = (emu_path_library if backend.name == "emu" else gpu_path_library)(request.param)
    return request.param


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py

    oracle_py.build()
    return oracle_py
