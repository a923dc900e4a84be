#!/bin/bash
# CPU sanitizer run: the C host layer and the oracle built with gcc -fsanitize=address,undefined,
# then the CPU tests that exercise them (list surgery of checkpoint / make_move / restore, the readers, the
# oracle against the goldens).  GPU sanitizers are not available on this pool; this runs anywhere.
set -euo pipefail
cd "$(dirname "$0")/.."
make -C mpmc_amd/csrc -s
make -C mpmc_amd/host -s asan
make -C oracle -s asan
export MPMC_HOST_LIB="$PWD/mpmc_amd/host/libmpmc_host_asan.so"
export MPMC_ORACLE_LIB="$PWD/oracle/libmpmc_oracle_asan.so"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1"
export UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
exec python -m pytest tests/test_host.py tests/test_oracle_golden.py -x -q -m "not gpu" "$@"
