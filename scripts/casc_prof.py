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

ds = sys.argv[1] if len(sys.argv) > 1 else "example_float_columns"
mib = sys.argv[2] if len(sys.argv) > 2 else "1024"
lib = nvcomp_amd.load_library()
import bench  # noqa: E402

sys.argv = ["bench.py", "--algo", "cascaded", "--dataset", ds, "--mib-per-gpu", mib, "--unique-mib", "32", "--steps", "3", "--warmup", "1",
            "--no-cpu-baseline", "--no-extras"]
slots = (C.c_ulonglong * 12)()
lib.nvcompAmdCascProfRead(slots, 12)  # clear
try:
    bench.main()
except SystemExit:
    pass
names = ["headers_plan", "run_pools", "values_unpack", "delta", "expand_inner", "expand_outer_to_memory", "expand_pool_marks", "copy_out"]
if lib.nvcompAmdCascProfRead(slots, 12) > 0:
    tot = float(sum(slots)) or 1.0
    print(json.dumps({"dataset": ds, "phase_share": {n: round(v / tot, 4) for n, v in zip(names, slots)}, "cycles_total": tot}))
