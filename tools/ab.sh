#!/bin/bash
# A/B of library builds on one GPU box: the default build and raytracer_amd/ab_<name>/librt_amd.so for each name given,
# each through tools/perf_probe.py at 1080p 1024 spp (best of 3).   tools/ab.sh [-s "scenes"] name...
# Build a variant in the build container first:
#   make -C raytracer_amd/csrc -j8 BUILD=build_ab/NAME OUT=../ab_NAME EXTRA="..." ../ab_NAME/librt_amd.so
set -o pipefail
SCENES="lucy teapot cornell_lucy dual"
if [ "$1" = "-s" ]; then SCENES="$2"; shift 2; fi
mkdir -p gpurun_out
for name in base "$@"; do
    lib=$PWD/raytracer_amd/librt_amd.so
    [ "$name" != base ] && lib=$PWD/raytracer_amd/ab_$name/librt_amd.so
    [ -f "$lib" ] || { echo "missing $lib"; exit 1; }
    echo "== $name"
    RT_AMD_LIB=$lib timeout -k 10 240 python3 -u tools/perf_probe.py --sn 32 --reps 3 $SCENES 2>&1 | grep -E "sn=32|^[a-z_:-]+:" | sed 's/render tree.*//' | paste - - || exit 1
done
