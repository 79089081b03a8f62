import sys, os, time, ctypes as C, numpy as np
sys.path.insert(0, ".")
import torch, nvcomp_amd
from nvcomp_amd import _lib, datasets
from oracle import oracle_py as oracle
oracle.build()
dev = nvcomp_amd.TorchDevice("cuda:0")
data = datasets.silesia_style(64 << 20, 0)
CH = 1 << 20
chunks = datasets.split_chunks(data, CH)
comp = [oracle.ref_lz4_compress(c, 12) for c in chunks]
for path in sys.argv[1:]:
    lib = _lib.declare(C.CDLL(os.path.abspath(path)))
    codec = nvcomp_amd.BatchedCodec(lib, dev, "LZ4")
    codec.decompress(comp, [c.size for c in chunks], canary=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs, act, st = codec.decompress(comp, [c.size for c in chunks], canary=False)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    ok = (st == 0).all() and all(np.array_equal(o, c) for o, c in zip(outs, chunks))
    print(os.path.basename(path), "64 chunks of 1 MiB: ok", ok, "wall incl. copies %.1f ms" % (t * 1e3))
