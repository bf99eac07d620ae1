#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the library's host-only C++ (ibo_amd/csrc/direct_host.cpp: DIRECT's tree logic), CPU only.
#   bash tools/sanitize_host.sh [out-file]
R=$(cd "$(dirname "$0")/.." && pwd)
B=$(mktemp -d)
g++ -std=c++17 -O1 -g -fno-omit-frame-pointer -DIBO_DIRECT_SELFCHECK -fsanitize=address,undefined -fno-sanitize-recover=undefined -I "$R/ibo_amd/csrc" \
    "$R/tools/direct_host_check.cpp" "$R/ibo_amd/csrc/direct_host.cpp" -o "$B/direct_host_check" || exit 2
ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 "$B/direct_host_check" 2>&1 | tee ${1:-/dev/null} | tail -5
rc=${PIPESTATUS[0]}
rm -rf "$B"
exit $rc
