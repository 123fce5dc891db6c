#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the HOST side of the library (config.cpp, host_setup.cpp, sim.cpp, link.cpp,
# rccl_link.cpp, recut.cpp: the .cpp files; the .hip files are compiled as usual -- GPU sanitizers are not available on this pool), then the
# whole CPU test suite against that build.  Runs in the build container, no GPU.  usage: scripts/host_sanitizers.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
mkdir -p "$TMP/opal_amd"; cp -r "$ROOT/opal_amd/csrc" "$TMP/opal_amd/csrc"; cp -r "$ROOT/include" "$TMP/include"
rm -f "$TMP"/opal_amd/csrc/*.o
make -s -C "$TMP/opal_amd/csrc" -j8 OUT="$TMP/libopal_asan.so" \
  HOSTFLAGS='-O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -ffp-contract=off -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -Wno-unused-value' \
  "$TMP/libopal_asan.so"
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
cd "$ROOT"
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 OPAL_AMD_LIB="$TMP/libopal_asan.so" \
  python -m pytest tests -x -q -m "not gpu"
# the parallel loader (host_setup.cpp: cell blocks on host threads) under ThreadSanitizer
rm -f "$TMP"/opal_amd/csrc/*.o
make -s -C "$TMP/opal_amd/csrc" -j8 OUT="$TMP/libopal_tsan.so" \
  HOSTFLAGS='-O1 -g -std=c++17 -fPIC -fsanitize=thread -fno-omit-frame-pointer -ffp-contract=off -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -Wno-unused-value' \
  "$TMP/libopal_tsan.so"
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.tsan-x86_64.so)
LD_PRELOAD=$RT TSAN_OPTIONS="report_signal_unsafe=0 exitcode=66" OPAL_AMD_LIB="$TMP/libopal_tsan.so" \
  python -m pytest tests/test_host_side.py -x -q -k loader
rm -rf "$TMP"
