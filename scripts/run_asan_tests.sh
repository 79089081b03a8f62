#!/bin/bash
# Kernel sources under AddressSanitizer: rebuild tests/emu with -fsanitize=address, run the kernel-level parity and
# fuzz tests with libasan preloaded into python, then restore the normal emulator build. CPU only.
set -u
cd "$(dirname "$0")/.."
make -C tests/emu clean > /dev/null
make -C tests/emu SAN="-fsanitize=address -fno-omit-frame-pointer" > /dev/null || exit 1
ASAN=$(gcc -print-file-name=libasan.so)
rm -f /tmp/nvcomp_asan.*
rc=0
for t in tests/test_fuzz_corrupt.py tests/test_fuzz_decode.py tests/test_bitcomp.py tests/test_ans.py tests/test_cascaded.py \
         tests/test_lz4_decode.py tests/test_lz4_encode.py tests/test_snappy.py tests/test_golden_decode.py tests/test_deflate.py tests/test_token_index.py; do
  LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:log_path=/tmp/nvcomp_asan \
    python -m pytest $t -x -q -m "not gpu" 2>&1 | grep -v "^Extension" | tail -1
done
if grep -l "ERROR: AddressSanitizer" /tmp/nvcomp_asan.* 2> /dev/null; then
  echo "AddressSanitizer reports above"; rc=1
else
  echo "no AddressSanitizer reports"
fi
make -C tests/emu clean > /dev/null
make -C tests/emu > /dev/null
exit $rc
