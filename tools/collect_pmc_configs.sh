#!/bin/bash
# PMC passes + kernel trace for the slow BASELINE configs at 1080p 1024 spp (run on the GPU box through gpurun).
# usage: tools/collect_pmc_configs.sh TAG SCENE...   -> gpurun_out/pmc_TAG/SCENE/{pmc_summary.txt,kernel_stats.csv}
set -u
TAG=$1; shift
R=$GRAFT_REPO_ROOT
for scene in "$@"; do
  OUT=gpurun_out/pmc_$TAG/$scene
  mkdir -p $R/$OUT
  cd $R && bash tools/prof_pmc.sh $OUT/pmc $scene 1920 1080 32 0 > $OUT/pmc.log 2>&1 || { echo "pmc $scene failed"; tail -5 $OUT/pmc.log; exit 1; }
  python3 tools/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.txt
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $R/$OUT/trace -o t --output-format csv -- python3 $R/tools/prof_render.py $scene 1920 1080 32 0 3 > $R/$OUT/trace.log 2>&1 || { echo "trace $scene failed"; exit 1; }
  find $R/$OUT/trace -name "*kernel_stats.csv" -exec cp {} $R/$OUT/kernel_stats.csv \;
  rm -rf $R/$OUT/trace $R/$OUT/pmc/*/  # keep the summaries only (raw csv is large)
  echo "$scene done"
done
