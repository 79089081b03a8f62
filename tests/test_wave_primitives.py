"""The wave64 primitives every kernel is built from (common/wave.h), on known inputs."""
import ctypes as C

import numpy as np


def test_wave_primitives(backend):
    d = backend.dev
    rng = np.random.RandomState(1)
    v = rng.randint(0, 1 << 20, size=64).astype(np.uint32)
    din = d.upload(v.view(np.uint8))
    dout = d.upload(np.zeros(640, dtype=np.uint32).view(np.uint8))
    scratch = d.upload(np.zeros(4096, dtype=np.uint8))
    fn = backend.lib.nvcompAmdSelfTestWave
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    fn.restype = C.c_int
    assert fn(d.ptr(din), d.ptr(dout), d.ptr(scratch), d.stream()) == 0
    d.synchronize()
    out = d.download(dout).view(np.uint32).reshape(10, 64)
    lanes = np.arange(64)
    assert np.array_equal(out[0], np.cumsum(v.astype(np.uint64)).astype(np.uint32))
    assert (out[1] == v.max()).all()
    assert (out[2] == np.uint32(v.astype(np.uint64).sum() & 0xFFFFFFFF)).all()
    assert np.array_equal(out[3], v[(lanes * 7 + 3) & 63])
    assert (out[4] == v[37]).all()
    ballot = sum(1 << i for i in range(64) if v[i] & 1)
    assert (out[5] == (ballot & 0xFFFFFFFF)).all() and (out[6] == (ballot >> 32)).all()
    exp = v.copy()
    exp[11] = 0xABCD
    assert np.array_equal(out[7], exp)
    assert np.array_equal(out[8], ((v + 1) & 0xFF)[::-1])
    assert (out[9] == 0).all()
