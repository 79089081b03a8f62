#!/usr/bin/env python3
"""Phase clock of the Cascaded decoder (a -DNVCOMP_CASC_PROF build, scripts/build_casc_variant.sh): share of wave cycles per
phase of casc::decompress_sub on the bench's float columns. usage: NVCOMP_AMD_LIB=<prof build> casc_prof.py [dataset] [mib]"""
import ctypes as C
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import nvcomp_amd  # noqa: E402

pos = [a for a in sys.argv[1:] if not a.startswith("--")]
ds = pos[0] if pos else "example_float_columns"
mib = pos[1] if len(pos) > 1 else "1024"
compress = "--compress" in sys.argv
lib = nvcomp_amd.load_library()
import bench  # noqa: E402

sys.argv = ["bench.py", "--algo", "cascaded", "--dataset", ds, "--mib-per-gpu", mib, "--unique-mib", "32", "--steps", "3", "--warmup", "1",
            "--no-cpu-baseline"] + ([] if compress else ["--no-extras"])
slots = (C.c_ulonglong * 12)()
lib.nvcompAmdCascProfRead(slots, 12)  # clear
try:
    bench.main()
except SystemExit:
    pass
names = ["headers_plan", "run_pools", "values_unpack", "delta", "expand_inner", "expand_outer_to_memory", "expand_pool_marks", "copy_out",
         "c_heads", "c_rle", "c_runs_packed", "c_stage_delta"]
if compress:  # slots 6 / 7 are shared with the compressor's last two phases
    names[6], names[7] = "c_values_ranged", "c_values_packed"
if lib.nvcompAmdCascProfRead(slots, 12) > 0:
    tot = float(sum(slots)) or 1.0
    print(json.dumps({"dataset": ds, "phase_share": {n: round(v / tot, 4) for n, v in zip(names, slots)}, "cycles_total": tot}))
