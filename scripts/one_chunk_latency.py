import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
import nvcomp_amd
from nvcomp_amd import datasets
from nvcomp_amd.batched import make_batch, empty_batch
from oracle import oracle_py as oracle
oracle.build()
lib = nvcomp_amd.load_library(); dev = nvcomp_amd.TorchDevice("cuda:0")
for fmt, enc in (("LZ4", oracle.ref_lz4_compress), ("Snappy", oracle.ref_snappy_compress)):
    codec = nvcomp_amd.BatchedCodec(lib, dev, fmt)
    for name in ("zeros", "noise", "text", "int32"):
        c = datasets.CLASSES[name](65536, 1)
        comp = make_batch(dev, [enc(c)], align=1)
        out = empty_batch(dev, [65536])
        actual = dev.upload(np.zeros(1, np.uint64).view(np.uint8)); st = dev.upload(np.zeros(1, np.int32).view(np.uint8))
        tb = codec.decompress_temp_size(1, 65536); temp = dev.empty(tb)
        for _ in range(3): codec.decompress_async(comp, out, actual, st, temp, tb)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): codec.decompress_async(comp, out, actual, st, temp, tb)
        e1.record(); torch.cuda.synchronize()
        ok = bool(np.array_equal(dev.download(out.slab)[:65536], c))
        print(fmt, name, "us per call %.1f" % (e0.elapsed_time(e1) / 50 * 1e3), ok, flush=True)
