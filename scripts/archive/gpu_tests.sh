#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-t}
mkdir -p "$OUT"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -15 "$OUT/pytest_gpu.log"
