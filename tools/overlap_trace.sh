#!/bin/bash
# Do two renders on two streams really run side by side?  Kernel start/end timestamps of tools/overlap_probe.py.
# usage: tools/overlap_trace.sh SCENE   -> gpurun_out/overlap_trace/
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/overlap_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python3 $R/tools/overlap_probe.py --frames 6 $1 > $OUT/probe.log 2>&1 || { tail -5 $OUT/probe.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "render_persistent" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows[-14:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    gap = "" if prev_end is None else f" start - previous end = {(s - prev_end) / 1e6:+.3f} ms"
    print(f"queue {r.get('Queue_Id', '?')} stream {r.get('Stream_Id', '?')}: start {s / 1e6:9.3f} ms, end {e / 1e6:9.3f} ms, {(e - s) / 1e6:7.3f} ms{gap}")
    prev_end = e
PY
