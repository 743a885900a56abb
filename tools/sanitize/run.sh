#!/bin/bash
# Builds librt_amd.so's host code with ThreadSanitizer, then with AddressSanitizer (+UBSan), into build_san/<kind>
# and runs the host-only scene-creation path under each.  CPU only: run in the build container.
#   tools/sanitize/run.sh [scene.yaml] [rounds]
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SCENE=${1:-$ROOT/scenes/lucy.yaml}
ROUNDS=${2:-3}
cd "$ROOT/raytracer_amd/csrc"
for kind in thread address; do
    out=$ROOT/build_san/$kind
    mkdir -p "$out"
    san="-Xarch_host -fsanitize=$kind"
    [ $kind = address ] && san="$san -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize=vptr"
    make -j8 BUILD="$out/obj" OUT="$out" EXTRA="-g -fno-omit-frame-pointer $san" LDEXTRA="-fsanitize=$kind -fno-gpu-sanitize" "$out/librt_amd.so" >"$out/build.log" 2>&1 \
        || { tail -20 "$out/build.log"; exit 1; }
    /opt/rocm/bin/hipcc -g -O1 -std=c++17 -fsanitize=$kind -fno-gpu-sanitize "$ROOT/tools/sanitize/create_host_only.cpp" \
        -L"$out" -lrt_amd -Wl,-rpath,"$out" -o "$out/create_host_only"
    echo "== $kind sanitizer: $SCENE x $ROUNDS"
    (cd "$ROOT" && TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1" ASAN_OPTIONS="detect_leaks=1:detect_stack_use_after_return=1" \
        "$out/create_host_only" "$SCENE" "$ROUNDS") 2>&1 | tee "$out/run.log" | tail -60
done
