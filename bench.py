#!/usr/bin/env python3
"""bench.py -- the hot path's headline metric on MI355X.

One "step" = one nvcompBatched<Algo>DecompressAsync call over the whole batch,
issued exactly as the reference's harness does it
(benchmarks/benchmark_template_chunked.cuh:519-536: events around a single
*Async call, statuses and actual sizes non-null) with the inputs already
resident in HBM. Throughput = uncompressed bytes / time
(benchmark_template_chunked.cuh:603-607).

Workload at N=1 = BASELINE.json configs[1]: "LZ4 batched decompress on
1xMI355X: CPU-compressed 64 KiB chunks": a Silesia-style synthetic mix
(nvcomp_amd/datasets.py), cut into 64 KiB chunks, compressed on the host with
liblz4's LZ4_compress_HC level 12 (the producer the reference's own example
uses, examples/lz4_cpu_compression.cu:61-66), `--unique-mib` MiB of unique data
replicated into distinct device memory up to `--mib-per-gpu` (the reference's
-x duplication, benchmark_template_chunked.cuh:340-353).

N>1: one process per GPU (torch.distributed / RCCL only for the barrier and the
max-over-ranks); every rank decodes its own shard of the batch, no data-path
collective: weak scaling. `--allgather` adds the benchmark_allgather.cpp
exchange (compress shard -> all-gather compressed bytes -> decode remote shards).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling is ~6300
CHUNK = 1 << 16


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--algo", choices=["lz4", "snappy", "cascaded", "bitcomp", "ans", "deflate"], default="lz4",
                   help="lz4 is the headline (BASELINE.json configs[1]); cascaded/bitcomp/ans are this library's own stream "
                        "formats: their inputs are made by the HIP compressor (checked against the CPU model) outside the timed region")
    p.add_argument("--opts", default="", help="cascaded: chunk_size,type,num_RLEs,num_deltas,use_bp; bitcomp: algo,type")
    p.add_argument("--mib-per-gpu", type=int, default=None,
                   help="uncompressed MiB decoded per GPU per step (default 4096; 1024 with --allgather, whose compaction "
                        "step builds a mask over every output slot)")
    p.add_argument("--unique-mib", type=int, default=64, help="unique MiB generated + CPU-compressed per rank")
    p.add_argument("--unique-kib", type=int, default=0, help="(tests) unique KiB per rank, overrides --unique-mib/--mib-per-gpu")
    p.add_argument("--dataset", default=None,
                   help="nvcomp_amd.datasets generator; default silesia_style (float_columns for cascaded / bitcomp: BASELINE.json configs[3])")
    p.add_argument("--producer", choices=["hc", "fast", "port"], default="hc",
                   help="CPU compressor making the inputs: liblz4 HC-12 / liblz4 default / oracle port")
    p.add_argument("--null-statuses", "--unchecked", dest="unchecked", action="store_true",
                   help="pass statuses = NULL (doc/lowlevel_c_quickstart.md:140): the same bounds-checked kernels, no report")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extras", action="store_true", help="skip the GPU compress leg (ratio / compress GB/s) and the riders")
    p.add_argument("--no-riders", action="store_true", help="skip the other codecs' lines in extras (the compress leg stays)")
    p.add_argument("--allgather", action="store_true", help="benchmark_allgather.cpp path (N >= 2)")
    p.add_argument("--no-verify", action="store_true", help="(profiling of ablated kernels only) skip output checks")
    p.add_argument("--dry-run-emu", action="store_true",
                   help="CPU-only self-test of this script's plumbing against tests/emu (prints value=null)")
    return p.parse_args()


def cpu_compress(oracle, algo, chunks, producer, threads):
    """Host-side producer of the compressed inputs (outside every timed region)."""
    caps = None
    if algo == "deflate":
        # the reference's CPU producers (examples/deflate_cpu_compression.cu:58-104): zlib deflateInit2(level 9, -15);
        # "fast" = level 1. Through the oracle/_ref shim's thread pool where it is built, else from Python threads.
        level = 1 if producer == "fast" else 9
        label = f"zlib deflate level {level}, raw streams (windowBits -15)"
        if oracle.have_ref():
            codec = oracle.ZLIB_DEFLATE_1 if level == 1 else oracle.ZLIB_DEFLATE_9
            _, outs, errs = oracle.batch_run(codec, chunks, [c.size + c.size // 8 + 64 for c in chunks], threads=threads, use_ref=True)
            assert errs == 0
            return [o.copy() for o in outs], label
        import zlib
        from multiprocessing.pool import ThreadPool

        def one(c):
            o = zlib.compressobj(level, zlib.DEFLATED, -15, 8)
            return np.frombuffer(o.compress(c.tobytes()) + o.flush(), dtype=np.uint8)

        with ThreadPool(threads) as pool:
            return pool.map(one, chunks), label
    if producer != "port" and oracle.have_ref():
        if algo == "lz4":
            codec = oracle.LZ4_ENC_HC if producer == "hc" else oracle.LZ4_ENC
            caps = [oracle.lz4_bound(c.size) + 64 for c in chunks]
        else:
            codec = oracle.SNAPPY_ENC
            caps = [oracle.snappy_bound(c.size) + 64 for c in chunks]
        _, outs, errs = oracle.batch_run(codec, chunks, caps, threads=threads, use_ref=True)
        assert errs == 0
        return [o.copy() for o in outs], ("liblz4 LZ4_compress_HC(12)" if codec == oracle.LZ4_ENC_HC else
                                          "liblz4 LZ4_compress_default" if algo == "lz4" else "libsnappy")
    codec = oracle.LZ4_ENC if algo == "lz4" else oracle.SNAPPY_ENC
    bound = oracle.lz4_bound if algo == "lz4" else oracle.snappy_bound
    _, outs, errs = oracle.batch_run(codec, chunks, [bound(c.size) for c in chunks], threads=threads)
    assert errs == 0
    return [o.copy() for o in outs], "oracle port (greedy)"


class TorchRuntime:
    """Streams, events, barrier and max-over-ranks on the real GPU(s)."""

    def __init__(self, torch, dist, dev):
        self.torch, self.dist, self.dev = torch, dist, dev

    def repeat(self, buf, k):
        return buf.repeat(k)

    def event(self):
        return self.torch.cuda.Event(enable_timing=True)

    def barrier_sync(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.dev.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def equal(self, a, b):
        return bool(self.torch.equal(a, b))

    def as_tensor(self, buf):
        return buf

    def side_streams(self, k):
        return [self.torch.cuda.Stream(device=self.dev.device) for _ in range(k)]

    def on_stream(self, s):
        return self.torch.cuda.stream(s)

    def wait_works(self, works):
        for w in works:
            w.wait()

    def join_streams(self, streams):
        """The current stream continues after everything queued on `streams`."""
        cur = self.torch.cuda.current_stream(self.dev.device)
        for s in streams:
            cur.wait_stream(s)

    def shutdown(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


class EmuRuntime:
    """--dry-run-emu only: wall-clock 'events', numpy buffers, gloo between ranks."""

    def __init__(self, dist=None):
        self.dist = dist

    class _Event:
        def record(self):
            self.t = time.perf_counter()

        def elapsed_time(self, other):
            return (other.t - self.t) * 1e3

    def repeat(self, buf, k):
        return np.tile(buf, k)

    def event(self):
        return EmuRuntime._Event()

    def barrier_sync(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        import torch

        t = torch.tensor([seconds], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def equal(self, a, b):
        return bool(np.array_equal(a, b))

    def as_tensor(self, buf):
        import torch

        return torch.from_numpy(buf)

    def side_streams(self, k):
        return [None] * k

    def on_stream(self, s):
        import contextlib

        return contextlib.nullcontext()

    def wait_works(self, works):
        # a gloo work may be waited for ONCE: a second wait() on a completed send / receive never returns (found by the
        # three-rank dry run: with two peers the second peer's "stream" waited for the same works again)
        if works is not getattr(self, "_waited", None):
            for w in works:
                w.wait()
            self._waited = works

    def join_streams(self, streams):
        pass

    def shutdown(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def setup_runtime(args):
    """Process-wide state: ranks, device, library. Returns a context dict."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one process per GPU (or none: bench.py launches itself)")
    import nvcomp_amd
    from nvcomp_amd import datasets
    from nvcomp_amd.batched import DeviceBatch

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.dry_run_emu:
        # plumbing self-test only: numpy "device", kernels compiled for the host (tests/emu)
        sys.path.insert(0, os.path.join(REPO, "tests"))
        import conftest as emu_conftest

        lib, dev = emu_conftest.emu_library(), emu_conftest.HostDevice()
        edist = None
        if world > 1 or os.environ.get("NVCOMP_AMD_BENCH_FORCE_DIST") == "1":  # CPU-only multi-process self-test: gloo
            import torch.distributed as edist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            edist.init_process_group("gloo", rank=rank, world_size=world)
        rt = EmuRuntime(edist)
    else:
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        # NVCOMP_AMD_BENCH_FORCE_DIST=1: bring RCCL up even for one rank, so a 1-GPU box can exercise the
        # barrier / max-over-ranks / digest-gather calls the N>1 launch relies on (tests/test_programs.py)
        use_dist = world > 1 or os.environ.get("NVCOMP_AMD_BENCH_FORCE_DIST") == "1"
        if use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        lib = nvcomp_amd.load_library()  # raises when the HIP library is missing
        dev = nvcomp_amd.TorchDevice(f"cuda:{local_rank}")
        rt = TorchRuntime(torch, dist if use_dist else None, dev)
    return {"rank": rank, "world": world, "lib": lib, "dev": dev, "rt": rt}


def run_case(args, ctx):
    """Build the batch, time `steps` decompress calls, verify, return the result dict (rank 0: full)."""
    import nvcomp_amd
    from nvcomp_amd import datasets
    from nvcomp_amd.batched import DeviceBatch

    rank, world, lib, dev, rt = ctx["rank"], ctx["world"], ctx["lib"], ctx["dev"], ctx["rt"]
    fmt = {"lz4": "LZ4", "snappy": "Snappy", "cascaded": "Cascaded", "bitcomp": "Bitcomp", "ans": "ANS",
           "deflate": "Deflate"}[args.algo]
    own_format = args.algo in OWN_FORMAT_OPTS
    opts = None
    if own_format:
        opts = tuple(int(x) for x in args.opts.split(",")) if args.opts else OWN_FORMAT_OPTS[args.algo]
    codec = nvcomp_amd.BatchedCodec(lib, dev, fmt, opts)
    # host threads of the (untimed) input producer and of the cpu_baseline leg: the box's cores shared among the ranks
    threads = max(1, len(os.sched_getaffinity(0)) // max(1, world))

    # ---- build the batch (untimed) ----
    from oracle import oracle_py as oracle  # producer of inputs + cpu_baseline checker only

    oracle.build()
    unique = (args.unique_kib << 10) if args.unique_kib else (args.unique_mib << 20)
    gen = getattr(datasets, args.dataset) if hasattr(datasets, args.dataset) else datasets.CLASSES[args.dataset]
    data = gen(unique, rank)
    chunks = datasets.split_chunks(data, CHUNK)
    if own_format:
        comp, producer = own_format_compress(oracle, codec, args.algo, opts, chunks)
    else:
        comp, producer = cpu_compress(oracle, args.algo, chunks, args.producer, threads)
    n_unique = len(chunks)
    replicas = 1 if args.unique_kib else max(1, (args.mib_per_gpu << 20) // unique)
    n = n_unique * replicas
    comp_sizes = np.array([c.size for c in comp], dtype=np.uint64)
    if own_format:  # Cascaded wants 4-byte aligned compressed chunks: pack on 8-byte boundaries
        padded = [np.concatenate([c, np.zeros((-c.size) % 8, dtype=np.uint8)]) for c in comp]
    else:  # tight-packed, unaligned (examples/BatchData.h:97-103)
        padded = comp
    pad_sizes = np.array([c.size for c in padded], dtype=np.uint64)
    comp_offs = np.zeros(n_unique, dtype=np.uint64)
    comp_offs[1:] = np.cumsum(pad_sizes)[:-1]
    comp_total = int(comp_sizes.sum())
    comp_host = np.concatenate(padded)
    slab_stride = int(pad_sizes.sum())
    raw_sizes = np.array([c.size for c in chunks], dtype=np.uint64)
    raw_offs = np.arange(n_unique, dtype=np.uint64) * np.uint64(CHUNK)

    comp_slab = rt.repeat(dev.upload(comp_host), replicas)  # distinct device memory per replica
    out_slab = dev.empty(unique * replicas)
    base_dev = dev.upload(data)

    def tile(offs, stride_bytes, base_ptr):
        rep = (np.arange(replicas, dtype=np.uint64) * np.uint64(stride_bytes))[:, None]
        return (offs[None, :] + rep + np.uint64(base_ptr)).reshape(-1)

    comp_batch = DeviceBatch(
        comp_slab, dev.upload(tile(comp_offs, slab_stride, dev.ptr(comp_slab)).view(np.uint8)),
        dev.upload(np.tile(comp_sizes, replicas).view(np.uint8)), None, np.tile(comp_sizes, replicas), n)
    out_batch = DeviceBatch(
        out_slab, dev.upload(tile(raw_offs, unique, dev.ptr(out_slab)).view(np.uint8)),
        dev.upload(np.tile(raw_sizes, replicas).view(np.uint8)), None, np.tile(raw_sizes, replicas), n)
    actual = dev.upload(np.zeros(n, dtype=np.uint64).view(np.uint8))
    statuses = None if args.unchecked else dev.upload(np.full(n, -1, dtype=np.int32).view(np.uint8))
    tb = codec.decompress_temp_size(n, CHUNK)
    temp = dev.empty(tb) if tb else None

    def step():
        rc = codec.decompress_async(comp_batch, out_batch, actual, statuses, temp, tb)
        if rc != 0:
            raise RuntimeError(f"nvcompBatched{fmt}DecompressAsync returned {rc}")

    # ---- timed region ----
    for _ in range(args.warmup):
        step()
    rt.barrier_sync()
    ev0, ev1 = rt.event(), rt.event()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    rt.barrier_sync()
    elapsed = rt.max_over_ranks(time.perf_counter() - t0)
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream

    # ---- verification (untimed): statuses, sizes, every output byte ----
    if args.no_verify:
        statuses, replicas_to_check = None, 0
    else:
        replicas_to_check = replicas
    if statuses is not None:
        st = dev.download(statuses).view(np.int32)[:n]
        if not (st == 0).all():
            # say whether the INPUT was at fault: the same compressed chunks through the CPU decoder (checker only)
            bad = np.nonzero(st != 0)[0]
            detail = []
            if not own_format and args.algo != "deflate":
                dec = oracle.ref_lz4_decompress if args.algo == "lz4" else oracle.ref_snappy_decompress
                for i in bad[:8]:
                    u = int(i) % n_unique
                    rc, ref = dec(comp[u], chunks[u].size) if oracle.have_ref() else (None, None)
                    detail.append((int(i), int(st[i]), "cpu decoder: " + ("ok" if rc == 0 and np.array_equal(ref, chunks[u]) else f"rc={rc}")))
            raise AssertionError(f"{bad.size} chunks failed: {detail}")
    act = dev.download(actual).view(np.uint64)[:n]
    assert args.no_verify or (act == np.tile(raw_sizes, replicas)).all(), "actual sizes differ from the originals"
    for r in range(replicas_to_check):
        assert rt.equal(out_slab[r * unique: (r + 1) * unique], base_dev[:unique]), f"replica {r} differs"

    total_raw = unique * replicas
    total_comp = comp_total * replicas
    value = world * total_raw * args.steps / elapsed / 1e9
    algorithmic = total_comp + total_raw + 44 * n  # C + U + 44 B of metadata per chunk
    achieved = algorithmic / (kernel_ms * 1e-3) / 1e9

    result = {
        "metric": f"decompress GB/s ({args.algo}, 64 KiB chunks)",
        "value": round(value, 3),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": (f"u{8 * _WIDTH[opts[1]]}" if args.algo in ("cascaded", "bitcomp") else "u8"),
        "data": "synthetic",
        "config": {
            "workload": (f"{fmt} batched decompress, 64 KiB chunks, inputs from this library's HIP compressor"
                         + (" (BASELINE.json configs[3])" if args.algo == "cascaded" else "")) if own_format else
                        f"{fmt} batched decompress, CPU-compressed 64 KiB chunks "
                        + ("(SURVEY.md 8 f4; the LZ4 line is BASELINE.json configs[1])" if args.algo == "deflate"
                           else "(BASELINE.json configs[1])"),
            "dataset": args.dataset,
            "producer": producer,
            "chunk_bytes": CHUNK,
            "chunks_per_gpu": n,
            "uncompressed_bytes_per_gpu": total_raw,
            "compressed_bytes_per_gpu": total_comp,
            "ratio": round(total_raw / total_comp, 4),
            "unique_bytes": unique,
            "statuses": "checked" if not args.unchecked else "null (same bounds-checked kernel, no per-chunk report)",
            "verified": not args.no_verify,
            "sharding": "chunks partitioned across ranks, no collective" if world > 1 else "single GPU",
        },
    }
    if rank == 0:
        kernel = (f"{args.algo}_decompress_kernel" if own_format or args.algo == "deflate" else lz_decode_kernel(args.algo, n))
        result["roofline"] = roofline_block(kernel, algorithmic, kernel_ms, args.algo, "decompress", args.dataset, n, producer)
    if rank == 0 and world == 1 and not args.no_extras:
        # compress leg on the GPU (ratio + compress GB/s of the metric string); not part of `value`
        from nvcomp_amd.batched import empty_batch

        max_out = codec.max_compressed_size(CHUNK)
        k = n
        src = DeviceBatch(out_slab, out_batch.ptrs, out_batch.sizes, None, out_batch.host_sizes[:k], k)
        dst = empty_batch(dev, [max_out] * k, stride=max_out)
        ctb = codec.compress_temp_size(k, CHUNK)
        ctemp = dev.empty(ctb) if ctb else None
        codec.compress_async(src, dst, CHUNK, ctemp, ctb)
        rt.barrier_sync()
        c0, c1 = rt.event(), rt.event()
        c0.record()
        for _ in range(5):
            codec.compress_async(src, dst, CHUNK, ctemp, ctb)
        c1.record()
        rt.barrier_sync()
        csz = dev.download(dst.sizes).view(np.uint64)[:k]
        raw_k = int(out_batch.host_sizes[:k].sum())
        comp_ms = c0.elapsed_time(c1) / 5
        comp_alg = raw_k + int(csz.sum()) + 40 * k  # U + C + 40 B of pointer/size traffic per chunk
        # the streams the compressor wrote, through the CPU library's decoder (the checker): a bounded sample
        sample = min(k, 256)
        csz_s = csz[:sample].astype(np.int64)
        host_c = dev.download(dst.slab, (sample - 1) * max_out + int(csz_s[-1]))
        bad = 0
        for i in range(sample):
            cc = host_c[i * max_out: i * max_out + int(csz_s[i])]
            orig = chunks[i % n_unique]
            if own_format:
                rc, back = _own_model(oracle, args.algo, opts)[1](cc, orig.size)
            elif args.algo == "deflate":
                import zlib
                rc, back = 0, np.frombuffer(zlib.decompress(cc.tobytes(), -15), dtype=np.uint8)
            else:
                dec = ((oracle.ref_lz4_decompress if args.algo == "lz4" else oracle.ref_snappy_decompress) if oracle.have_ref()
                       else (oracle.lz4_decompress if args.algo == "lz4" else oracle.snappy_decompress))
                rc, back = dec(cc, orig.size)
            bad += int(rc != 0 or not np.array_equal(back, orig))
        assert bad == 0, f"{bad} of {sample} GPU-compressed chunks are not restored by the CPU decoder"
        checked_by = (f"oracle/{args.algo}_ref.c (CPU model of this library's stream)" if own_format else
                      "zlib inflate" if args.algo == "deflate" else
                      ("liblz4 LZ4_decompress_safe" if args.algo == "lz4" else "snappy::RawUncompress") if oracle.have_ref()
                      else "oracle/ C port")
        kname = (f"{args.algo}_compress_wide_kernel" if args.algo in ("lz4", "snappy") else f"{args.algo}_compress_kernel")
        result["extras"] = {
            "gpu_compress_GBps": round(raw_k / (comp_ms * 1e-3) / 1e9, 3),
            "gpu_compress_ratio": round(raw_k / int(csz.sum()), 4),
            "gpu_compress_chunks": k,
            "gpu_compress_checked_by": f"{checked_by}: {sample} chunks of the timed launch's output, bit-exact",
            "compress_roofline": roofline_block(kname, comp_alg, comp_ms, args.algo, "compress", args.dataset, k),
        }
        if args.algo == "deflate":
            # algo 0 above: the fixed Huffman code; algo 1: per-chunk codes (two runs of the match finder + code construction)
            dyn = nvcomp_amd.BatchedCodec(lib, dev, fmt, (1,))
            dyn.compress_async(src, dst, CHUNK, ctemp, ctb)
            rt.barrier_sync()
            d0, d1 = rt.event(), rt.event()
            d0.record()
            for _ in range(3):
                dyn.compress_async(src, dst, CHUNK, ctemp, ctb)
            d1.record()
            rt.barrier_sync()
            dsz = dev.download(dst.sizes).view(np.uint64)[:k]
            result["extras"]["gpu_compress_dynamic_codes"] = {
                "GBps": round(raw_k / (d0.elapsed_time(d1) / 3 * 1e-3) / 1e9, 3), "ratio": round(raw_k / int(dsz.sum()), 4)}
        del dst, ctemp
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # The reference's CPU path (liblz4 / snappy decoders) on this box's host cores over a
        # bounded sample of the same chunk arrays: the unique set, best of 5.
        if own_format:
            result["cpu_baseline"] = own_format_cpu_baseline(oracle, args.algo, comp, chunks, threads)
            return finish(result, args, world, rt, data)
        if args.algo == "deflate":
            result["cpu_baseline"] = deflate_cpu_baseline(oracle, comp, chunks, threads, unique)
            return finish(result, args, world, rt, data)
        use_ref = oracle.have_ref()
        code = oracle.LZ4_DEC if args.algo == "lz4" else oracle.SNAPPY_DEC
        # bounded sample: the unique set repeated so that every thread gets >= 16 MiB per run (thread start-up
        # dominates a small sample on a many-core host) but at most 1 GiB in all, best of 5 runs. (On the 2 x EPYC
        # 9575F box a 4 GiB sample measures 53 GB/s where 1 GiB measures ~130: the larger output falls out of the
        # 768 MB of L3. The figure kept is the one that favours the CPU.)
        reps = max(1, min(replicas, 1024 // max(1, args.unique_mib),
                          (16 * threads + args.unique_mib - 1) // max(1, args.unique_mib)))
        s_comp, s_caps = comp * reps, [c.size for c in chunks] * reps
        secs, outs, errs = oracle.batch_run(code, s_comp, s_caps, threads=threads, repeats=5, use_ref=use_ref)
        assert errs == 0 and all(o.size == c for o, c in zip(outs, s_caps))
        result["cpu_baseline"] = {
            "value": round(unique * reps / secs / 1e9, 3),
            "unit": "GB/s",
            "cores": threads,
            "kind": "reference" if use_ref else "port",
            "sample": f"{(unique * reps) >> 20} MiB ({n_unique * reps} chunks) of the same workload, best of 5, "
                      + ("liblz4 LZ4_decompress_safe" if (use_ref and args.algo == "lz4") else
                         "libsnappy RawUncompress" if use_ref else "oracle/ C port"),
        }
        # the CPU peer of the compress leg: the fast compressor of the same library on the same
        # sample. (libdeflate, which north_star also names, is the CPU peer of the DEFLATE rider: extras.deflate.cpu_baseline.)
        enc = oracle.LZ4_ENC if args.algo == "lz4" else oracle.SNAPPY_ENC
        bound = (CHUNK + CHUNK // 255 + 16) if args.algo == "lz4" else (32 + CHUNK + CHUNK // 6)
        s_raw = chunks * reps
        csecs, couts, cerrs = oracle.batch_run(enc, s_raw, [bound] * len(s_raw), threads=threads, repeats=3, use_ref=use_ref)
        if cerrs == 0:
            result["cpu_baseline"]["compress"] = {
                "value": round(unique * reps / csecs / 1e9, 3), "unit": "GB/s", "cores": threads,
                "ratio": round(unique * reps / max(1, sum(int(o.size) for o in couts)), 4),
                "kind": "reference" if use_ref else "port",
                "sample": ("liblz4 LZ4_compress_default" if args.algo == "lz4" else "libsnappy RawCompress") if use_ref
                          else "oracle/ C port"}
    return finish(result, args, world, rt, data)


def deflate_cpu_baseline(oracle, comp, chunks, threads, unique):
    """The CPU peers of the DEFLATE path on this box's host cores, one C thread per core through the oracle/_ref shim, over
    the unique set repeated until every thread has a few dozen chunks (best of 3): libdeflate_deflate_decompress -- the
    peer BASELINE.json's north_star names and the reference's algo 0 (examples/deflate_cpu_compression.cu:60-67,
    deflate_cpu_decompression.cu) -- is `value`; zlib inflate (algo 1/2, examples/deflate_cpu_decompression.cu:128-170)
    rides beside it; the compress peers: libdeflate level 6 (as the reference calls it) and zlib level 1."""
    import zlib

    reps = max(1, min(16, (32 * threads) // max(1, len(comp))))
    total = unique * reps
    if oracle.have_ref():
        s_comp, s_caps = comp * reps, [c.size for c in chunks] * reps
        secs, outs, errs = oracle.batch_run(oracle.ZLIB_INFLATE, s_comp, s_caps, threads=threads, repeats=3, use_ref=True)
        assert errs == 0 and all(o.size == c for o, c in zip(outs, s_caps))
        csecs, couts, cerrs = oracle.batch_run(oracle.ZLIB_DEFLATE_1, chunks * reps, [c + c // 8 + 64 for c in s_caps],
                                               threads=threads, repeats=1, use_ref=True)
        assert cerrs == 0
        sample = f"{total >> 20} MiB ({len(s_comp)} chunks) of the same workload, best of 3, one thread per core"
        zl = {"value": round(total / secs / 1e9, 3), "unit": "GB/s", "cores": threads, "kind": "reference",
              "sample": f"{sample}, zlib {zlib.ZLIB_VERSION} inflate (raw streams)",
              "compress": {"value": round(total / csecs / 1e9, 3), "unit": "GB/s", "cores": threads,
                           "ratio": round(total / max(1, sum(int(o.size) for o in couts)), 4), "kind": "reference",
                           "sample": "zlib deflate level 1"}}
        if not oracle.have_libdeflate():
            return zl
        lsecs, louts, lerrs = oracle.batch_run(oracle.LIBDEFLATE_DEC, s_comp, s_caps, threads=threads, repeats=3, use_ref=True)
        assert lerrs == 0 and all(o.size == c for o, c in zip(louts, s_caps))
        assert all(np.array_equal(o, c) for o, c in zip(louts[: len(chunks): 53], chunks[::53]))
        esecs, eouts, eerrs = oracle.batch_run(oracle.LIBDEFLATE_ENC_6, chunks * reps, [c + c // 8 + 64 for c in s_caps],
                                               threads=threads, repeats=2, use_ref=True)
        assert eerrs == 0
        return {"value": round(total / lsecs / 1e9, 3), "unit": "GB/s", "cores": threads, "kind": "reference",
                "sample": f"{sample}, libdeflate_deflate_decompress (raw streams)",
                "compress": {"value": round(total / esecs / 1e9, 3), "unit": "GB/s", "cores": threads,
                             "ratio": round(total / max(1, sum(int(o.size) for o in eouts)), 4), "kind": "reference",
                             "sample": "libdeflate_deflate_compress level 6 (examples/deflate_cpu_compression.cu:62)"},
                "zlib": zl}
    from multiprocessing.pool import ThreadPool

    blobs = [c.tobytes() for c in comp] * reps

    def one(b):
        return len(zlib.decompress(b, -15))

    best = None
    with ThreadPool(min(threads, 64)) as pool:
        for _ in range(3):
            t0 = time.perf_counter()
            sizes = pool.map(one, blobs, chunksize=max(1, len(blobs) // (8 * min(threads, 64))))
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
    assert sizes == [c.size for c in chunks] * reps
    return {"value": round(total / best / 1e9, 3), "unit": "GB/s", "cores": min(threads, 64), "kind": "reference",
            "sample": f"{total >> 20} MiB ({len(blobs)} chunks) of the same workload, best of 3, zlib {zlib.ZLIB_VERSION} inflate "
                      "from Python threads"}


def lz_decode_kernel(algo, chunks):
    """The kernel nvcompBatched{LZ4,Snappy}DecompressAsync launches for a batch of this size (compile-time thresholds of
    common/lz_launch.hip.h: a workgroup per chunk / two waves per chunk / persistent waves)."""
    if chunks <= 512:  # NVCOMP_LZ_TEAM_MAX_BATCH (tests/test_abi.py keeps the two in step)
        return f"{algo}_decompress_team_kernel"
    return f"{algo}_decompress_pair_kernel" if chunks <= 4096 else f"{algo}_decompress_window_kernel"  # NVCOMP_LZ_PAIR_MAX_BATCH


PMC_RECORD = "pmc_traffic_r06.json"
# Said ONCE per line (`notes`), not in every roofline object: the driver keeps the last 8 KB of stdout.
NOTES = {
    "traffic": "fabric bytes at the L2's memory side = 2 x FETCH_SIZE + WRITE_SIZE (every L2 miss is a 128-byte request tallied "
               "at 64, profiles/r04_feasibility.json); Infinity-Cache hits are such requests too: an upper bound of the DRAM bytes",
    "traffic_src": "r = replayed from profiles/" + PMC_RECORD + " (separate rocprofv3 --pmc passes of this kernel, this workload and "
                   "these kernel sources: keyed to a digest of the sources); stale = recorded for another build, not replayed; "
                   "none = no record",
    "issue": "valu_busy = SQ_INSTS_VALU x 4.1 cycles / (1 024 SIMDs x kernel_ms x sclk), salu_busy = SQ_INSTS_SALU x 4.17 / the same: "
             "cycles per wave64 instruction per SIMD MEASURED on this card (scripts/microbench/valu_issue.hip -> "
             "profiles/r06_valu_issue_{a,b,c,d}.jsonl: 4.06-4.17 for every mix of vector operations, 2.2 only for pure streams of "
             "add / sub / logic / mov / right shifts; scalar 4.17); sclk = 2.3 GHz, what s_memtime / s_memrealtime and the "
             "PMC passes (SQ_BUSY_CYCLES) show under these kernels, not the 2.4 GHz of the data sheet",
}
VALU_CYCLES, SALU_CYCLES = 4.1, 4.17  # profiles/r06_valu_issue_*.jsonl
SCLK_HZ = 2.3e9  # measured under load (profiles/r06_valu_issue_*.jsonl: memtime_ghz 2.2-2.4; PMC pass of the decoder: 2.24)


def replayed_counters(algo, kind, dataset, chunks, producer=None):
    """Counters are PMC measurements (separate rocprofv3 --pmc passes, scripts/gpu_traffic.sh): they cannot be taken
    inside this run, so the committed record is REPLAYED -- only for the same kernel, the same workload AND the same
    kernel sources; otherwise null. Returns (record or None, "r" | "stale" | "none": NOTES["traffic_src"])."""
    path = os.path.join(REPO, "profiles", PMC_RECORD)
    if not os.path.exists(path):
        return None, "none"
    try:
        records = json.load(open(path))
    except Exception:
        return None, "none"
    digest = library_source_digest(algo)
    stale = False
    for rec in records:
        if rec.get("algo") == algo and rec.get("kind") == kind and rec.get("dataset") == dataset and rec.get("chunks_per_gpu") == chunks:
            if kind == "decompress" and producer is not None and rec.get("producer") not in (None, producer):
                continue  # the same data through another compressor is another stream (round 6: the sorted-key column, HC and default)
            if rec.get("lib_source_digest") == digest:
                return rec, "r"
            stale = True
    return None, ("stale" if stale else "none")


def roofline_block(kernel, algorithmic, kernel_ms, algo, kind, dataset, chunks, producer=None):
    """The `roofline` object of a line: useful bytes against the HBM peak, the replayed fabric traffic, and the issue side
    (for the LZ kernels it is vector issue, not bytes, that binds). What the fields mean is said once, in `notes`."""
    achieved = algorithmic / (kernel_ms * 1e-3) / 1e9
    rec, source = replayed_counters(algo, kind, dataset, chunks, producer)
    block = {
        "bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 4), "algorithmic_bytes_per_launch": int(algorithmic),
        "kernel_ms": round(kernel_ms, 4), "traffic": rec.get("hbm_bytes_per_launch") if rec else None,
        "traffic_x": round(rec["hbm_bytes_per_launch"] / algorithmic, 2) if rec and rec.get("hbm_bytes_per_launch") else None,
        "traffic_src": source,
    }
    if rec and rec.get("valu_wave_insts"):
        simd_cycles = 1024 * kernel_ms * 1e-3 * SCLK_HZ
        block["issue"] = {
            "valu": int(rec["valu_wave_insts"]), "salu": int(rec.get("salu_wave_insts") or 0),
            "valu_busy": round(rec["valu_wave_insts"] * VALU_CYCLES / simd_cycles, 3),
            "salu_busy": round((rec.get("salu_wave_insts") or 0) * SALU_CYCLES / simd_cycles, 3),
        }
    return block


def summary_of(result):
    """The last object of the line: every BASELINE.json config in a few hundred bytes -- GB/s (of uncompressed bytes), the
    kernel's roofline fraction and its traffic multiple, both directions where the config is a round trip."""
    def leg(value, roof, ratio=None):
        if value is None:
            return None
        d = {"GBps": round(value, 1), "frac": roof.get("frac"), "traffic_x": roof.get("traffic_x")}
        if ratio is not None:
            d["ratio"] = ratio
        return d

    ex = result.get("extras", {})
    out = {}
    name = result["metric"].split("(")[1].split(",")[0]
    out[f"{name}_dec"] = leg(result["value"], result.get("roofline", {}))
    if "gpu_compress_GBps" in ex:
        out[f"{name}_comp"] = leg(ex["gpu_compress_GBps"], ex.get("compress_roofline", {}), ex.get("gpu_compress_ratio"))
    for key, val in ex.items():
        if not isinstance(val, dict) or "value" not in val:
            continue
        if val.get("value") is None:
            out[f"{key}_dec"] = {"error": val.get("error", "")[:80]}
            continue
        out[f"{key}_dec"] = leg(val["value"], val.get("roofline", {}))
        if "chunks_per_gpu" in val:
            out[f"{key}_dec"]["chunks"] = val["chunks_per_gpu"]
        if isinstance(val.get("compress"), dict):
            c = val["compress"]
            out[f"{key}_comp"] = leg(c["value"], c.get("roofline", {}), c.get("ratio"))
    if "cpu_baseline" in result:
        cb = result["cpu_baseline"]
        out["cpu_dec"] = {"GBps": cb.get("value"), "cores": cb.get("cores")}
        if isinstance(cb.get("compress"), dict):
            out["cpu_comp"] = {"GBps": cb["compress"].get("value"), "ratio": cb["compress"].get("ratio")}
    return out


def library_source_digest(algo="lz4"):
    """sha256 over the kernel sources of one codec: PMC traffic recorded for one build of a kernel must not be replayed
    beside the timing of another. LZ4 / Snappy: their own directories + common/ (the shared
    decoder) + their api/*_api.hip (the kernels' launch shapes); DEFLATE likewise; the own formats: their own directory."""
    import glob
    import hashlib

    dirs = ["lz4", "snappy", "common"] if algo in ("lz4", "snappy") else ["deflate", "common"] if algo == "deflate" else [algo]
    h = hashlib.sha256()
    paths = [p for d in dirs for p in sorted(glob.glob(os.path.join(REPO, "nvcomp_amd", "csrc", d, "*.h*")))]
    if algo in ("lz4", "snappy", "deflate"):
        paths += [os.path.join(REPO, "nvcomp_amd", "csrc", "api", f"{a}_api.hip")
                  for a in (("lz4", "snappy") if algo in ("lz4", "snappy") else ("deflate",))]
    for path in paths:
        h.update(os.path.relpath(path, REPO).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def finish(result, args, world, rt, data):
    if rt.dist is not None:
        digests = [None] * world
        rt.dist.all_gather_object(digests, shard_digest(data))
        result["config"]["shard_digests"] = digests
    if args.dry_run_emu:
        result["value"] = None
        result["data"] = "DRY RUN on the CPU emulation of the kernels: not a measurement"
    return result


# default options of the own-format codecs (cascaded: benchmark_cascaded_chunked.cu:35-36 run with -t int)
OWN_FORMAT_OPTS = {"cascaded": (4096, 4, 2, 1, 1), "bitcomp": (0, 4), "ans": (0,)}
_WIDTH = [1, 1, 2, 2, 4, 4, 8, 8]


def _own_model(oracle, algo, opts):
    if algo == "cascaded":
        return (lambda c: oracle.cascaded_compress(c, *opts)), oracle.cascaded_decompress
    if algo == "bitcomp":
        return (lambda c: oracle.bitcomp_compress(c, opts[0], _WIDTH[opts[1]])), oracle.bitcomp_decompress
    return oracle.ans_compress, oracle.ans_decompress


def own_format_compress(oracle, codec, algo, opts, chunks):
    """Inputs of an own-format decode run: the HIP compressor's output (outside every timed region); the first
    chunks must equal the CPU model's bytes and decode with it."""
    comp = []
    for i in range(0, len(chunks), 512):
        comp.extend(codec.compress(chunks[i: i + 512]))
    enc, dec = _own_model(oracle, algo, opts)
    for c, cc in list(zip(chunks, comp))[:4]:
        assert np.array_equal(cc, enc(c)), "HIP compressor output differs from the CPU model"
        rc, back = dec(cc, c.size)
        assert rc == 0 and np.array_equal(back, c)
    return comp, f"this library's HIP {algo} compressor, opts {tuple(opts)}"


def own_format_cpu_baseline(oracle, algo, comp, chunks, threads):
    """The CPU model of the stream (oracle/*_ref.c: a scalar port, written for clarity) over a bounded sample, one
    chunk per task on `threads` host threads (oracle/batch.c), best of 3 runs."""
    code = {"cascaded": oracle.CASCADED_DEC, "bitcomp": oracle.BITCOMP_DEC, "ans": oracle.ANS_DEC}[algo]
    # >= 64 chunks (4 MiB) per thread, at most 16 x the unique set (1 GiB by default): starting and joining 256 threads
    # costs milliseconds, a smaller sample would time that instead of the codec
    reps = max(1, min(16, (64 * threads + len(comp) - 1) // max(1, len(comp))))
    s_comp, s_caps = list(comp) * reps, [c.size for c in chunks] * reps
    secs, outs, errs = oracle.batch_run(code, s_comp, s_caps, threads=threads, repeats=3)
    assert errs == 0 and all(o.size == c for o, c in zip(outs, s_caps))
    assert all(np.array_equal(o, c) for o, c in zip(outs[: len(chunks): 97], chunks[::97]))
    raw = sum(s_caps)
    return {"value": round(raw / secs / 1e9, 3), "unit": "GB/s", "cores": threads, "kind": "port",
            "sample": f"{raw >> 20} MiB ({len(s_caps)} chunks) of the same workload, best of 3, oracle/{algo}_ref.c "
                      "(scalar CPU model of this library's own stream; the reference has no CPU implementation of it)"}


def shard_digest(data):
    """Cheap content fingerprint of a rank's shard (shows that ranks work on different chunks)."""
    import zlib

    return zlib.crc32(np.ascontiguousarray(data[: 1 << 20]).tobytes()) & 0xFFFFFFFF


def run_allgather_case(args, ctx):
    """benchmark_allgather.cpp semantics over RCCL (reference: benchmarks/benchmark_allgather.cpp:288-470):
    every rank LZ4-compresses its shard on its GPU, the compressed bytes are all-gathered over xGMI
    (two collectives per step: the chunk sizes, then the compacted payloads padded to the largest
    rank's total), every rank decompresses the G-1 remote shards and ends up with all the data.
    Unlike the reference, the payload moved is the ACTUAL compressed size, not the bound."""
    import torch

    import nvcomp_amd
    from nvcomp_amd import datasets
    from nvcomp_amd.batched import DeviceBatch

    rank, world, lib, dev, rt = ctx["rank"], ctx["world"], ctx["lib"], ctx["dev"], ctx["rt"]
    dist = rt.dist
    forced = os.environ.get("NVCOMP_AMD_BENCH_FORCE_DIST") == "1"  # one rank: collectives run, nothing is remote
    assert dist is not None and (world >= 2 or forced), "--allgather needs at least 2 ranks"
    codec = nvcomp_amd.BatchedCodec(lib, dev, "LZ4")
    unique = (args.unique_kib << 10) if args.unique_kib else (args.unique_mib << 20)
    gen = getattr(datasets, args.dataset) if hasattr(datasets, args.dataset) else datasets.CLASSES[args.dataset]
    data = gen(unique, rank)
    replicas = 1 if args.unique_kib else max(1, (args.mib_per_gpu << 20) // unique)
    shard_bytes = unique * replicas
    n = shard_bytes // CHUNK
    raw = rt.as_tensor(rt.repeat(dev.upload(data), replicas))
    max_out = (codec.max_compressed_size(CHUNK) + 7) // 8 * 8
    slots = rt.as_tensor(dev.empty(n * max_out))
    sizes = rt.as_tensor(dev.upload(np.zeros(n, dtype=np.int64).view(np.uint8))).view(torch.int64)
    raw_sizes = np.full(n, CHUNK, dtype=np.uint64)

    def ptr(t):
        return int(t.data_ptr())

    def dev_u64(arr):
        return dev.upload(np.asarray(arr, dtype=np.uint64).view(np.uint8))

    src = DeviceBatch(raw, dev_u64(ptr(raw) + np.arange(n, dtype=np.uint64) * CHUNK), dev_u64(raw_sizes), None, raw_sizes, n)
    dst = DeviceBatch(slots, dev_u64(ptr(slots) + np.arange(n, dtype=np.uint64) * max_out), sizes, None, raw_sizes, n)
    ctb = codec.compress_temp_size(n, CHUNK)
    ctemp = dev.empty(ctb) if ctb else None
    dtb = codec.decompress_temp_size(n, CHUNK)
    out = rt.as_tensor(dev.empty(shard_bytes * world))
    remote = [r for r in range(world) if r != rank]
    m = n * len(remote)
    # Every buffer of a step exists before the timed loop: the packed payload of this rank and one
    # receive buffer per peer at the worst-case size (288 GB of HBM: 8 ranks x 4 GiB shards need 7 x 4.02 GiB), the
    # gathered sizes and the pointer arrays derived from them.
    cap_bytes = n * max_out
    packed = rt.as_tensor(dev.empty(cap_bytes))
    my_offsets = rt.as_tensor(dev.upload(np.zeros(n + 1, dtype=np.int64).view(np.uint8))).view(torch.int64)
    recv = {r: rt.as_tensor(dev.empty(cap_bytes)) for r in remote}
    all_sizes_flat = torch.zeros(world * n, dtype=torch.int64, device=sizes.device)
    all_sizes = all_sizes_flat.view(world, n)
    ptrs_all = torch.zeros(world, n, dtype=torch.int64, device=sizes.device)
    actual = rt.as_tensor(dev.upload(np.zeros(max(m, 1), dtype=np.uint64).view(np.uint8))).view(torch.int64)
    statuses = rt.as_tensor(dev.upload(np.full(max(m, 1), -1, dtype=np.int32).view(np.uint8))).view(torch.int32)
    dtemps = {r: (dev.empty(dtb) if dtb else None) for r in remote}
    out_batches = {}
    for r in remote:
        optrs = ptr(out) + r * shard_bytes + np.arange(n, dtype=np.uint64) * CHUNK
        out_batches[r] = DeviceBatch(out, dev_u64(optrs), dev_u64(np.full(n, CHUNK)), None, None, n)
    side = rt.side_streams(len(remote))
    moved = [0]
    # The exchange runs in SLICES of the chunk range: slice j of every shard travels in ONE grouped send/recv
    # (ncclGroupStart .. ncclSend/ncclRecv to and from every peer .. ncclGroupEnd: every xGMI link of the rank carries its
    # own peer's bytes at once, no ring, no root) and is decoded, peer by peer on the peers' streams, while slice j+1 is on
    # the wire (the reference waits for the whole exchange, benchmark_allgather.cpp:370).
    n_slices = max(1, min(4, n // 4096)) if not args.unique_kib else min(2, n)
    cuts = [n * j // n_slices for j in range(n_slices + 1)]
    cut_index = torch.tensor(cuts, dtype=torch.int64, device=sizes.device)
    cut_offs = torch.zeros(world, n_slices + 1, dtype=torch.int64, device=sizes.device)  # byte offset of every cut, per rank
    offs_ext = torch.zeros(world, n + 1, dtype=torch.int64, device=sizes.device)  # exclusive prefix sums + the total

    def step():
        rc = codec.compress_async(src, dst, CHUNK, ctemp, ctb)
        assert rc == 0, rc
        # the chunks leave their worst-case slots for one contiguous payload: device-side prefix sum + one wave per chunk
        rc = lib.nvcompAmdBatchedPackAsync(dev.ptr(dst.ptrs), dev.ptr(sizes), n, dev.ptr(packed), cap_bytes,
                                           dev.ptr(my_offsets), dev.stream())
        assert rc == 0, rc
        dist.all_gather_into_tensor(all_sizes_flat, sizes)
        torch.cumsum(all_sizes, dim=1, out=offs_ext[:, 1:])
        torch.index_select(offs_ext, 1, cut_index, out=cut_offs)
        host_cuts = cut_offs.tolist()  # THE host sync of a step: byte counts, like the reference's sync_all_streams (:370)
        for row in host_cuts:  # every peer's cuts: ascending and inside the receive buffers
            assert row[0] == 0 and all(a <= b for a, b in zip(row, row[1:])) and row[-1] <= cap_bytes, row
        moved[0] = int(sum(row[-1] for row in host_cuts))
        for j in range(n_slices):
            c0, c1 = cuts[j], cuts[j + 1]
            ops = []
            for r in remote:  # the ACTUAL compressed bytes travel, not the bound the reference ships
                ops.append(dist.P2POp(dist.irecv, recv[r][host_cuts[r][j]: host_cuts[r][j + 1]], r))
                ops.append(dist.P2POp(dist.isend, packed[host_cuts[rank][j]: host_cuts[rank][j + 1]], r))
            works = dist.batch_isend_irecv(ops) if ops else []
            for k, r in enumerate(remote):
                with rt.on_stream(side[k]):
                    rt.wait_works(works)  # RCCL: this stream waits for the exchange; gloo: the host does, once
                    torch.add(offs_ext[r, c0:c1], ptr(recv[r]), out=ptrs_all[r, c0:c1])
                    batch = DeviceBatch(recv[r], ptrs_all[r, c0:c1], all_sizes[r, c0:c1], None, None, c1 - c0)
                    ob = out_batches[r]
                    oslice = DeviceBatch(out, ob.ptrs[8 * c0: 8 * c1], ob.sizes[8 * c0: 8 * c1], None, None, c1 - c0)
                    rc = codec.decompress_async(batch, oslice, actual[k * n + c0: k * n + c1],
                                                statuses[k * n + c0: k * n + c1], dtemps[r], dtb)
                    assert rc == 0, rc
        out[rank * shard_bytes: (rank + 1) * shard_bytes].copy_(raw)  # own shard: plain copy (benchmark_allgather.cpp:386-393)
        rt.join_streams(side)

    for _ in range(args.warmup):
        step()
    rt.barrier_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    rt.barrier_sync()
    elapsed = rt.max_over_ranks(time.perf_counter() - t0)
    if m:
        st = statuses[:m].cpu().numpy()
        assert (st == 0).all(), f"{int((st != 0).sum())} remote chunks failed"
    # every rank must now hold every shard: compare fingerprints with the owners'
    # (position-weighted byte sums, 64 MiB at a time: a 4 GiB shard as ONE int64 tensor is 32 GB and a launch the runtime
    # refuses -- found by tests/test_programs.py::test_bench_under_launcher_with_rccl[allgather_4gib] in round 4)
    piece = 64 << 20
    weights = (torch.arange(min(piece, shard_bytes), device=sizes.device) % 65521).to(torch.int64)

    def fingerprint(buf):
        acc = torch.zeros((), dtype=torch.int64, device=sizes.device)
        for at in range(0, shard_bytes, piece):
            part = buf[at: at + piece].to(torch.int64)
            acc += (part * weights[: part.numel()]).sum() * (1 + at // piece)
        return acc

    prints = torch.stack([fingerprint(out[r * shard_bytes: (r + 1) * shard_bytes]) for r in range(world)])
    owner = torch.zeros(world, dtype=torch.int64, device=sizes.device)
    own = fingerprint(raw).reshape(1)
    dist.all_gather_into_tensor(owner, own)
    assert torch.equal(prints, owner), "a rank holds wrong data after the all-gather"
    total = shard_bytes * world
    per_step = elapsed / args.steps
    return {
        "metric": "lz4 all-gather system GB/s (benchmark_allgather semantics)",
        "value": round(total * (world - 1) / per_step / 1e9, 3),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(per_step * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": "LZ4 compress shard -> RCCL all-gather of compressed bytes -> decompress remote shards "
                        "(BASELINE.json configs[4], benchmarks/benchmark_allgather.cpp)",
            "dataset": args.dataset,
            "uncompressed_bytes_per_gpu": shard_bytes,
            "chunks_per_gpu": n,
            "per_gpu_GBps": round(total * (world - 1) / world / per_step / 1e9, 3),
            "compressed_bytes_moved_per_step": moved[0],
            "ratio": round(total / max(1, moved[0]), 4),
        },
    }


def rider(args, ctx, algo, **overrides):
    """Another codec's decompress line on the same kind of workload, for the `extras` of the driver's line. Never lets
    a failure of its own spoil that line: the error text takes the place of the numbers."""
    import copy

    sargs = copy.copy(args)
    sargs.algo, sargs.no_extras, sargs.no_cpu_baseline, sargs.steps, sargs.warmup = algo, True, True, 5, 1
    sargs.opts = ""
    compress_leg = overrides.pop("compress_leg", False)  # BASELINE.json configs[2] / [3] are round trips
    sargs.no_extras = not compress_leg
    for key, val in overrides.items():
        setattr(sargs, key, val)
    try:
        r = run_case(sargs, ctx)
        line = {"value": r["value"], "unit": "GB/s", "ms_per_step": r["ms_per_step"], "roofline": r["roofline"],
                "ratio": r["config"]["ratio"], "producer": r["config"]["producer"], "verified": r["config"]["verified"],
                "chunks_per_gpu": r["config"]["chunks_per_gpu"], "dataset": r["config"]["dataset"]}
        if compress_leg and "extras" in r:
            e = r["extras"]
            line["compress"] = {"value": e["gpu_compress_GBps"], "unit": "GB/s", "ratio": e["gpu_compress_ratio"],
                                "chunks_per_gpu": e["gpu_compress_chunks"], "roofline": e["compress_roofline"],
                                "checked_by": e.get("gpu_compress_checked_by")}
        if "cpu_baseline" in r:
            line["cpu_baseline"] = r["cpu_baseline"]
        return line
    except Exception as e:  # noqa: BLE001 -- reported, not raised: see the docstring
        return {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}


def self_launch(args):
    """`python bench.py --gpus N` alone (no launcher, no WORLD_SIZE): become `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` -- one process per GPU,
    rank 0 prints the one JSON line. The reference's multi-GPU program is one command too
    (benchmarks/benchmark_allgather.cpp:594-644, -g N)."""
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    args = parse_args()
    if os.environ.get("NVCOMP_AMD_BENCH_WATCHDOG"):
        # a rank that hangs (a collective nobody answers) prints every thread's stack and exits instead of waiting for the
        # launcher's timeout: NVCOMP_AMD_BENCH_WATCHDOG=<seconds>
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["NVCOMP_AMD_BENCH_WATCHDOG"]), exit=True)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    if args.mib_per_gpu is None:
        args.mib_per_gpu = 4096
    if args.dataset is None:
        # BASELINE.json configs[3]: "int32 columnar floats" -> float columns shaped like the reference's ExampleFloatData.csv
        # BASELINE.json configs[3]: the reference's own ExampleFloatData.csv columns after text_to_binary.py
        args.dataset = ("example_float_columns" if args.algo == "cascaded" else "float_columns" if args.algo == "bitcomp"
                        else "silesia_style")
    ctx = setup_runtime(args)
    result = run_allgather_case(args, ctx) if args.allgather else run_case(args, ctx)
    if (ctx["rank"] == 0 and ctx["world"] == 1 and args.algo == "lz4" and not args.allgather and not args.no_extras
            and not args.no_riders and not args.dry_run_emu):
        # north_star bars BOTH LZ decoders: the Snappy line of the same workload rides along (5 timed launches); so
        # does the DEFLATE decoder's, on a quarter of the workload
        # (BASELINE.json configs[2] is a Snappy compress + decompress round trip: the compress leg rides with it)
        result.setdefault("extras", {})["snappy"] = rider(args, ctx, "snappy", compress_leg=True)
        # ... with its CPU peers timed beside it: libdeflate (the one north_star names) and zlib
        result["extras"]["deflate"] = rider(args, ctx, "deflate", mib_per_gpu=min(args.mib_per_gpu, 1024),
                                            unique_mib=min(args.unique_mib, 32), no_cpu_baseline=args.no_cpu_baseline)
        # BASELINE.json configs[3]: Cascaded {4096, int, 2 RLE, 1 delta, bit-packing} on the reference's own float columns
        # (benchmarks/benchmark_cascaded_chunked.cu:35-36), both directions
        result["extras"]["cascaded"] = rider(args, ctx, "cascaded", dataset="example_float_columns", compress_leg=True,
                                             mib_per_gpu=min(args.mib_per_gpu, 1024), unique_mib=min(args.unique_mib, 32))
        # the entropy coders (own streams), both directions, 1 GiB each
        result["extras"]["ans"] = rider(args, ctx, "ans", dataset="silesia_style", compress_leg=True,
                                        mib_per_gpu=min(args.mib_per_gpu, 1024), unique_mib=min(args.unique_mib, 32))
        result["extras"]["bitcomp"] = rider(args, ctx, "bitcomp", dataset="float_columns", compress_leg=True,
                                            mib_per_gpu=min(args.mib_per_gpu, 1024), unique_mib=min(args.unique_mib, 32))
        # the one shape the reference publishes a number for (doc/Benchmarks.md:88-95: LZ4 on Mortgage 2009Q2 column 0,
        # ratio 38.9, A100 decompress 320.7 GB/s): long matches and runs -- the data that CAN approach the roofline
        result["extras"]["lz4_mortgage_like"] = rider(args, ctx, "lz4", dataset="mortgage_col0_like", compress_leg=True,
                                                      mib_per_gpu=min(args.mib_per_gpu, 1024), unique_mib=min(args.unique_mib, 64))
        # ... the same column through liblz4's DEFAULT compressor (every run starts with a 6-byte match from an earlier key:
        # the run executor's speculated matches, common/lz_window.hip.h), and an int32 column (runs of period 4, one in twenty
        # shorter than 16 bytes)
        result["extras"]["lz4_mortgage_like_default"] = rider(args, ctx, "lz4", dataset="mortgage_col0_like", producer="fast",
                                                              mib_per_gpu=min(args.mib_per_gpu, 1024), unique_mib=min(args.unique_mib, 64))
        result["extras"]["lz4_int32"] = rider(args, ctx, "lz4", dataset="int32", producer="fast",
                                              mib_per_gpu=min(args.mib_per_gpu, 1024), unique_mib=min(args.unique_mib, 32))
        # ... and the sorted-key column at the batch size of the reference's published run (5 021 chunks: 5 120 here)
        if args.mib_per_gpu > 320:
            result["extras"]["lz4_mortgage_like_5120"] = rider(args, ctx, "lz4", dataset="mortgage_col0_like", mib_per_gpu=320,
                                                               unique_mib=min(args.unique_mib, 32))
        # The headline batch is the size where the tail of the last round of persistent waves vanishes; the reference's own
        # programs run 1 ... 8 192 chunks (benchmarks/benchmark_lz4_synth.cpp:64-72) and 5 021 (doc/Benchmarks.md:88-95):
        # the same mix, producer and checks at 16 384, 4 096 and 256 chunks (persistent waves / two waves per chunk / a
        # workgroup per chunk)
        if args.mib_per_gpu > 1024:
            result["extras"]["lz4_16384"] = rider(args, ctx, "lz4", mib_per_gpu=1024)
            result["extras"]["lz4_4096"] = rider(args, ctx, "lz4", mib_per_gpu=256)
            result["extras"]["lz4_256"] = rider(args, ctx, "lz4", mib_per_gpu=16, unique_mib=min(args.unique_mib, 16))
    if args.dry_run_emu and args.allgather:
        result["value"] = None
        result["data"] = "DRY RUN on the CPU emulation of the kernels: not a measurement"
    if ctx["rank"] == 0:
        if "roofline" in result:
            # said once; `summary` is the LAST object of the line: the driver's 8 KB tail always holds every config
            result["notes"] = NOTES
            cb = result.pop("cpu_baseline", None)
            if cb is not None:
                result["cpu_baseline"] = cb  # (behind the extras, in front of the summary)
            result["summary"] = summary_of(result)
        print(json.dumps(result), flush=True)
    if os.environ.get("NVCOMP_AMD_PROF"):  # phase clocks of a -DNVCOMP_LZW_PROF build (scripts/build_variants.sh)
        import ctypes
        import nvcomp_amd
        lib = nvcomp_amd.load_library()
        slots = (ctypes.c_ulonglong * 20)()
        names = ["in_ensure", "chase_build", "chase_enum", "parse", "exec_prep", "make_room", "literals_4_32",
                 "far_store", "match_rounds", "flush", "loop_top", "far_issue", "literals_1_3", "literals_long",
                 "matches_whole_wave", "token_index", "run_patterns", "run_joints", "run_bodies", "run_restart"]
        reader = "nvcompAmdProfReadSnappy" if args.algo == "snappy" else "nvcompAmdProfRead"
        if args.algo == "snappy":
            names[15] = "copy_trains"
        if args.algo == "deflate":  # scripts/build_deflate_variant.sh dprof -DNVCOMP_LZW_PROF: the front end's phases + the executor's
            reader = "nvcompAmdProfReadDeflate"
            for i, n in ((0, "headers_and_code_tables"), (1, "window_tables"), (2, "enumerations"), (3, "round_decode_and_records"),
                         (10, "symbols_one_at_a_time"), (15, "front_end_rest")):
                names[i] = n
        if hasattr(lib, reader) and getattr(lib, reader)(slots, 20) > 0:
            tot = float(sum(slots)) or 1.0
            print(json.dumps({"phase_share": {n: round(v / tot, 4) for n, v in zip(names, slots)},
                              "cycles_total": tot}), file=sys.stderr, flush=True)
    ctx["rt"].shutdown()


if __name__ == "__main__":
    main()
