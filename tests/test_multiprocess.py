"""N>1 path of bench.py on CPU: two (and eight) processes, gloo, the kernels' host emulation.

Checks what the real multi-GPU run relies on: rendezvous on 127.0.0.1, every rank decodes
its OWN shard (different content per rank), barrier + max-over-ranks timing, one JSON line
from rank 0; and for --allgather, that every rank ends up holding every shard
(the program asserts the fingerprints)."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def launch(extra, port=None, world=2):
    """port=None: `python bench.py --gpus 2 ...` exactly as the driver types it -- the program launches its own ranks
    (bench.self_launch); with a port: under the launcher, one process per rank, as the driver's N>1 contract spells it."""
    tail = ["--gpus", str(world), "--dry-run-emu", "--unique-kib", "128", "--steps", "1", "--warmup", "0"] + extra
    if port is None:
        cmd = [sys.executable, "bench.py"] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
               "127.0.0.1", "--master-port", str(port), "bench.py"] + tail
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    return json.loads(lines[0])


def test_sharded_decompress_two_ranks():
    res = launch(["--no-cpu-baseline", "--no-extras"])
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["value"] is None
    digests = res["config"]["shard_digests"]
    assert len(digests) == 2 and digests[0] != digests[1], "ranks must work on different shards"
    assert res["config"]["chunks_per_gpu"] == 2


def test_sharded_deflate_two_ranks():
    """The same path for the DEFLATE decoder: the chunks partition across ranks exactly like LZ4's."""
    res = launch(["--no-cpu-baseline", "--no-extras", "--algo", "deflate"], 29624)
    assert res["n_gpus"] == 2 and "deflate" in res["metric"]
    digests = res["config"]["shard_digests"]
    assert len(digests) == 2 and digests[0] != digests[1]


def test_allgather_two_ranks():
    res = launch(["--allgather"])
    assert res["n_gpus"] == 2 and "all-gather" in res["metric"]
    assert res["config"]["compressed_bytes_moved_per_step"] > 0
    assert res["config"]["ratio"] > 1.0


def test_allgather_step_is_library_code():
    """The timed step of --allgather allocates nothing and compacts with the library (it used to be
    boolean-mask indexing, torch.zeros/empty per step and two host syncs)."""
    import inspect
    import re

    sys.path.insert(0, REPO)
    import bench

    src = inspect.getsource(bench.run_allgather_case)
    step = src[src.index("    def step():"): src.index("    for _ in range(args.warmup):")]
    for banned in ("torch.zeros", "torch.empty", "dev.empty", "dev.upload", ".item()", "col <"):
        assert banned not in step, banned
    assert "nvcompAmdBatchedPackAsync" in step and "batch_isend_irecv" in step and "on_stream" in step
    assert "broadcast" not in step, "the payloads travel peer to peer in one grouped exchange, not rank by rank"
    assert len(re.findall(r"\.tolist\(\)|\.cpu\(\)", step)) == 1, "exactly one host sync: the sizes"


def test_sharded_decompress_eight_ranks():
    """The driver's widest run (N = 8, one process per GPU): eight shards, eight digests, one line."""
    res = launch(["--no-cpu-baseline", "--no-extras"], world=8)
    assert res["n_gpus"] == 8 and res["scaling"] == "weak"
    digests = res["config"]["shard_digests"]
    assert len(digests) == 8 and len(set(digests)) == 8, "eight ranks, eight different shards"
    assert res["config"]["chunks_per_gpu"] == 2


def test_allgather_eight_ranks():
    """benchmark_allgather.cpp's shape at the node's width: seven peers per rank -- seven receive buffers, seven side
    streams, seven slices of `actual` / `statuses`, the [world, slices + 1] table of byte cuts, and the fingerprint
    gather over eight owners (the program asserts all of it; world size 2 exercises one peer only)."""
    res = launch(["--allgather"], 29631, world=8)
    assert res["n_gpus"] == 8 and "all-gather" in res["metric"]
    cfg = res["config"]
    assert cfg["chunks_per_gpu"] == 2 and cfg["uncompressed_bytes_per_gpu"] == 128 << 10
    # every rank sends its packed shard: the eight payloads together, and they are compressed
    assert 0 < cfg["compressed_bytes_moved_per_step"] < 8 * (128 << 10)
