#!/bin/bash
# Round profile set for the bench workload (run on the GPU box through gpurun):
#   1. rocprofv3 --kernel-trace --stats over the very bench.py command line  -> kernel_stats.csv
#   2. PMC passes (one rocprofv3 run each) over one render call of the same workload -> pmc summary
# usage: tools/collect_profiles.sh TAG   (writes gpurun_out/prof_TAG/...)
set -u
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/trace -o bench --output-format csv -- \
    python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/bench_under_rocprof.err || { echo "kernel-trace run failed"; tail -5 $R/$OUT/bench_under_rocprof.err; exit 1; }
find $R/$OUT/trace -name "*kernel_stats.csv" -exec cp {} $R/$OUT/kernel_stats.csv \;
cd $R && bash tools/prof_pmc.sh $OUT/pmc lucy 1920 1080 32 0 > $OUT/pmc.log 2>&1 || { echo "pmc failed"; tail -5 $OUT/pmc.log; exit 1; }
python3 tools/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.txt
echo "profiles collected in $OUT"
cat $OUT/kernel_stats.csv
