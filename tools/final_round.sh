#!/bin/bash
# Everything the round's tables and profiles are made of, in one GPU call (run through gpurun from the repo root).
# usage: tools/final_round.sh TAG   -> gpurun_out/final_TAG/*
set -u
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=gpurun_out/final_$TAG
mkdir -p $R/$OUT
cd $R
step() { echo "== $*" >&2; "$@"; }
step bash tools/collect_profiles.sh $TAG > $OUT/collect_profiles.log 2>&1 || { tail -5 $OUT/collect_profiles.log; exit 1; }
step bash tools/collect_pmc_configs.sh $TAG teapot cornell_lucy dual > $OUT/collect_pmc_configs.log 2>&1 || { tail -5 $OUT/collect_pmc_configs.log; exit 1; }
step timeout -k 10 300 python3 -u tools/scene_table.py > $OUT/scene_table.md 2> $OUT/scene_table.err || { tail -5 $OUT/scene_table.err; exit 1; }
step timeout -k 10 300 python3 -u tools/example_table.py > $OUT/example_table.md 2> $OUT/example_table.err || { tail -5 $OUT/example_table.err; exit 1; }
step timeout -k 10 300 python3 -u tests/parity_table.py > $OUT/parity_ids.md 2> $OUT/parity_ids.err || { tail -5 $OUT/parity_ids.err; exit 1; }
step timeout -k 10 300 python3 -u tools/perf_probe.py --stages --stats-sn 16 --sn 16 lucy teapot cornell_lucy dual distant > $OUT/stage_split.txt 2>&1 || { tail -5 $OUT/stage_split.txt; exit 1; }
step timeout -k 10 300 python3 -u tools/overlap_probe.py lucy teapot cornell_lucy > $OUT/overlap.txt 2>&1 || { tail -5 $OUT/overlap.txt; exit 1; }
step timeout -k 10 300 python3 -u tools/build_probe.py > $OUT/build_probe.txt 2>&1 || { tail -5 $OUT/build_probe.txt; exit 1; }
step bash tools/run_configs.sh $OUT/configs.md > $OUT/run_configs.log 2>&1 || { tail -5 $OUT/run_configs.log; exit 1; }
echo "final round set in $OUT"
