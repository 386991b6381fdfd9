#!/bin/bash
# CPU test suite against the AddressSanitizer + UndefinedBehaviorSanitizer build of the oracle (make -C oracle asan).
# GPU sanitizers are not available on this pool; this covers the C code that runs on the host: the checker itself.
# usage (build container, repo root): bash tools/run_sanitized.sh [pytest args]
set -eu
cd "$(dirname "$0")/.."
make -s -C oracle asan
ASAN_RT=$(gcc -print-file-name=libasan.so)
# python itself is not instrumented: leak reports of the interpreter are noise, everything else aborts the run
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export D3D_ORACLE_SANITIZED=1
LD_PRELOAD=$ASAN_RT python3 -m pytest tests/test_oracle_voxel.py tests/test_oracle_box.py tests/test_point.py -q -m "not gpu" "$@"
