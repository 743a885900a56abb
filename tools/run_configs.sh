#!/bin/bash
# BASELINE.json configs C2-C5 through the CLI on one GPU (C4/C5 in full: 4K, 4096 / 16384 spp).
# usage: tools/run_configs.sh OUT.md
set -u
OUT=$1
R=$GRAFT_REPO_ROOT
BIN=$R/raytracer_amd/raytracer
mkdir -p /tmp/cfg
{
echo "| config | scene | size | spp | samples | kernel ms | Msamples/s | wall s (whole CLI run) |"
echo "|---|---|---|---|---|---|---|---|"
run() {  # name scene width height spp
    local j
    j=$(cd $R && timeout -k 10 400 $BIN --scene scenes/$2.yaml --width $3 --height $4 -n $5 --no-progress --json -o /tmp/cfg/$1_$2.png 2>/tmp/cfg/$1_$2.err | tail -1) || { echo "| $1 | $2 | failed |"; cat /tmp/cfg/$1_$2.err >&2; return 1; }
    python3 - "$1" "$2" "$3" "$4" "$5" "$j" <<'PY'
import json, sys
name, scene, w, h, spp, j = sys.argv[1:]
d = json.loads(j)
print(f"| {name} | {scene}.yaml | {w}x{h} | {spp} | {d['samples']:.4g} | {d['kernel_ms']:.1f} | {d['msamples_per_s']:.0f} | {d['wall_s']:.2f} |")
PY
}
run C1 cornell 400 400 100 &&
run C2 teapot 1920 1080 1024 &&
run C3 lucy 1920 1080 1024 &&
run C4 cornell_lucy 3840 2160 4096 &&
run C5 dual 3840 2160 16384 &&
run C5 distant 3840 2160 16384
} > $R/$OUT
cat $R/$OUT
