#!/bin/bash
# A/B over values of one environment knob of the library:  tools/env_ab.sh VAR "v1 v2 ..." [scenes...]
set -o pipefail
VAR=$1; VALUES=$2; shift 2
SCENES=${@:-lucy teapot cornell_lucy dual}
for v in $VALUES; do
    echo "== $VAR=$v"
    env $VAR=$v timeout -k 10 240 python3 -u tools/perf_probe.py --sn 32 --reps 3 --stats-sn 8 $SCENES 2>&1 | grep -E "sn=32|seg/sample" | paste - - | sed 's/  */ /g' || exit 1
done
