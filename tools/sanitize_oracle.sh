#!/bin/bash
# The CPU checker itself under AddressSanitizer + UndefinedBehaviorSanitizer: oracle/ibo_oracle.c built with -fsanitize=address,undefined and the
# oracle-vs-golden and oracle-vs-compiled-reference tests run against that build (CPU only; test infrastructure, never the product).
#   bash tools/sanitize_oracle.sh [out-file]
R=$(cd "$(dirname "$0")/.." && pwd)
B=$(mktemp -d)
gcc -O1 -g -fno-omit-frame-pointer -fopenmp -fPIC -shared -std=c99 -fsanitize=address,undefined -fno-sanitize-recover=undefined \
    -o "$B/liboracle_san.so" "$R/oracle/ibo_oracle.c" -lm || exit 2
ASAN=$(gcc -print-file-name=libasan.so)
cd "$R" && IBO_ORACLE_LIB="$B/liboracle_san.so" LD_PRELOAD="$ASAN" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 \
    timeout 1800 python3 -m pytest tests/test_oracle_golden.py tests/test_cpu_host.py -q -x -p no:cacheprovider -k "not sanitizers" 2>&1 | tee ${1:-/dev/null} | tail -5
rc=${PIPESTATUS[0]}
rm -rf "$B"
exit $rc
