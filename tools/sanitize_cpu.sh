#!/bin/bash
# AddressSanitizer / UBSan pass over the CPU-side code (GPU sanitizers are not available on the pool):
#   1. the oracle (gcc -fsanitize=address,undefined) under its own CPU tests,
#   2. the product's host code -- OBJ loader, BVH builder, photon-map balance, ABI glue -- (clang -fsanitize=address,
#      host pass only; the device objects of the normal build are linked unchanged) under the CPU tests that reach it.
# Instrumented libraries are built in a temporary directory and swapped in for the duration of the run.
# usage: bash tools/sanitize_cpu.sh      (from the repo root, after `make` in oracle/ and cse168-raytracer_amd/)
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d /tmp/miro_asan.XXXXXX)
PKG=$ROOT/cse168-raytracer_amd
restore() {
    [ -f "$T/orig_oracle.so" ] && cp "$T/orig_oracle.so" "$ROOT/oracle/libmiro_oracle.so"
    [ -f "$T/orig_hip.so" ] && cp "$T/orig_hip.so" "$PKG/lib/libmiro_hip.so"
    rm -rf "$T"
}
trap restore EXIT

echo "== oracle under ASan + UBSan"
for f in miro_oracle miro_oracle_shade miro_oracle_photon miro_oracle_path; do
    gcc -std=c99 -O1 -g -fPIC -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -I"$ROOT/oracle" -I"$ROOT/include" \
        -c "$ROOT/oracle/$f.c" -o "$T/$f.o"
done
gcc -std=gnu11 -O1 -g -fPIC -msse4.1 -fopenmp -ffp-contract=off -fsanitize=address,undefined -I"$ROOT/oracle" \
    -c "$ROOT/oracle/miro_oracle_sse.c" -o "$T/sse.o"
gcc -shared -fsanitize=address,undefined -fopenmp -o "$T/libmiro_oracle.so" "$T"/miro_oracle*.o "$T/sse.o" -lm
cp "$ROOT/oracle/libmiro_oracle.so" "$T/orig_oracle.so"
cp "$T/libmiro_oracle.so" "$ROOT/oracle/libmiro_oracle.so"
(cd "$ROOT" && LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
    ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    python -m pytest tests/test_oracle_kat.py tests/test_objects.py tests/test_photon.py tests/test_path_rays.py -x -q -m "not gpu" -p no:cacheprovider)
cp "$T/orig_oracle.so" "$ROOT/oracle/libmiro_oracle.so"

echo "== product host code under ASan"
for f in mr_api mr_obj mr_build mr_photon; do
    /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC -ffp-contract=off -fsanitize=address -shared-libasan --cuda-host-only \
        -I"$ROOT/include" -I"$PKG/csrc" -c "$PKG/csrc/$f.cpp" -o "$T/p_$f.o"
done
/opt/rocm/bin/hipcc -shared -fPIC -fsanitize=address -shared-libasan --offload-arch=gfx950 -o "$T/libmiro_hip.so" \
    "$T"/p_*.o "$PKG"/build/*.hip.o
cp "$PKG/lib/libmiro_hip.so" "$T/orig_hip.so"
cp "$T/libmiro_hip.so" "$PKG/lib/libmiro_hip.so"
(cd "$ROOT" && LD_PRELOAD="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so)" \
    ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:protect_shadow_gap=0 \
    python -m pytest tests/test_host_parity.py tests/test_abi.py tests/test_objects.py tests/test_photon.py -x -q -m "not gpu" -p no:cacheprovider)
echo "== clean"
