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


_emu_gather_lib = None


def emu_gather_library():
    """The same emulation build with the LZ decoders' alternative (byte-gather) batch executor compiled in."""
    global _emu_gather_lib
    if _emu_gather_lib is None:
        from nvcomp_amd import _lib

        emu_dir = _make_emu("gather")
        _emu_gather_lib = _lib.declare(C.CDLL(os.path.join(emu_dir, "libnvcomp_emu_gather.so")))
    return _emu_gather_lib


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
def emu_gather():
    return Backend("emu", emu_gather_library(), HostDevice())


@pytest.fixture(scope="session")
def gpu():
    import nvcomp_amd

    return Backend("gpu", nvcomp_amd.load_library(), nvcomp_amd.TorchDevice("cuda:0"))


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def backend(request):
    """Every test through this fixture runs the LZ4 decoder on its two-kernel path (token index + indexed decoder:
    opt-in in the library, include/nvcomp/amd_ext.h). Tests of the LZ decoders additionally ask for `lz_path` to see
    the single-kernel chase decoder (the default for large batches) and the two-waves-per-chunk decoder (the default
    for small ones) too."""
    b = request.getfixturevalue(request.param)
    b.lib.nvcompAmdSetLZPairMaxBatch(0)
    b.lib.nvcompAmdSetLZIndexMinBatch(1)
    return b


@pytest.fixture(params=["indexed", "chase", "pair"])
def lz_path(request, backend):
    """All decode paths of nvcompBatchedLZ4DecompressAsync, whatever the batch size."""
    backend.lib.nvcompAmdSetLZPairMaxBatch((1 << 60) if request.param == "pair" else 0)
    backend.lib.nvcompAmdSetLZIndexMinBatch(1 if request.param == "indexed" else 1 << 60)
    yield request.param
    backend.lib.nvcompAmdSetLZPairMaxBatch(0)
    backend.lib.nvcompAmdSetLZIndexMinBatch(1)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py

    oracle_py.build()
    return oracle_py
