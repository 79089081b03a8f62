#!/usr/bin/env python3
"""Statistics of the one-wave LZ decoders on the headline workload, counted on the host emulation of the kernel sources
(tests/emu, the build that forces the one-wave path): sequences, batches, far / near matches and near-match rounds per
batch, LZ4 (liblz4 HC-12 streams) beside Snappy (libsnappy streams). CPU only; writes profiles/r05_match_rounds.json.
VERDICT r4 item 5 asked for `mrr_rounds` beside the Snappy figure."""
import ctypes as C
import json
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def child(fmt):
    import numpy as np

    import nvcomp_amd
    from conftest import HostDevice, emu_path_library
    from nvcomp_amd import datasets
    from oracle import oracle_py as oracle

    oracle.build()
    lib = emu_path_library("chase")
    dev = HostDevice()
    codec = nvcomp_amd.BatchedCodec(lib, dev, fmt)
    enc = (lambda c: oracle.ref_lz4_compress(c, 12)) if fmt == "LZ4" else oracle.ref_snappy_compress
    data = datasets.silesia_style(64 * 65536, 0)
    chunks = datasets.split_chunks(data)
    comp = [enc(c) for c in chunks]
    outs, actual, status = codec.decompress(comp, [c.size for c in chunks])
    assert (status == 0).all() and all(np.array_equal(o, c) for o, c in zip(outs, chunks))
    C.CDLL(os.path.join(REPO, "tests", "emu", "libnvcomp_emu_chase.so")).emu_stats_dump(0)
    print("chunks", len(chunks), file=sys.stderr)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    out = {}
    for fmt in ("LZ4", "Snappy"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", fmt], capture_output=True, text=True, cwd=REPO)
        assert r.returncode == 0, r.stderr[-2000:]
        st = {m.group(1): int(m.group(2)) for m in re.finditer(r"stat (\S+)\s+(\d+)", r.stderr)}
        n = int(re.search(r"chunks (\d+)", r.stderr).group(1))
        b = max(1, st.get("batches", 1))
        out[fmt.lower()] = {
            "chunks": n, "sequences_per_chunk": round(st.get("seqs", 0) / n, 1), "batches_per_chunk": round(b / n, 1),
            "sequences_per_batch": round(st.get("seqs", 0) / b, 1),
            "far_matches_per_batch": round(st.get("match_far_lanes", 0) / b, 1),
            "near_matches_per_batch": round(st.get("match_near_lanes", 0) / b, 1),
            "whole_wave_matches_per_batch": round(st.get("match_coop", 0) / b, 2),
            "near_match_rounds_per_batch (mrr_rounds)": round(st.get("mrr_rounds", 0) / b, 2),
            "raw": st,
        }
    out["note"] = ("64 chunks of the Silesia-style mix (the bench line's generator, seed 0); emulator statistics of the one-wave "
                   "window decoder (LZ_STAT hooks in common/lz_window.hip.h), not timings")
    json.dump(out, open(os.path.join(REPO, "profiles", "r05_match_rounds.json"), "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "raw"} for k, v in out.items() if k != "note"}, indent=1))


if __name__ == "__main__":
    main()
