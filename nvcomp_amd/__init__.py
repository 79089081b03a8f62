"""nvcomp_amd -- MI355X-native batched lossless compression behind nvCOMP's API.

The product is the C-ABI shared library ``nvcomp_amd/lib/libnvcomp.so`` (HIP
kernels for gfx950 + the ``nvcompBatched*`` entry points declared in
``include/nvcomp/*.h``). This Python package is only the host-side plumbing used
by the tests and ``bench.py``: it loads the library with ctypes and hands it
device pointers of torch tensors. It contains no codec logic and no CPU
fallback: if the HIP library is missing, loading fails loudly.
"""
from ._lib import LIB_PATH, build_library, load_library, NvcompStatus, NvcompType  # noqa: F401
from .batched import BatchedCodec, TorchDevice  # noqa: F401

__version__ = "0.1.0"
