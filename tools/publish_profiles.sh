#!/bin/bash
# Copies what tools/final_round.sh TAG left under gpurun_out/ into profiles/ under the round's names, and writes the
# PMC record bench.py reads.   usage: tools/publish_profiles.sh TAG COMMIT [ROUND]    (in the build container)
set -eu
TAG=$1; COMMIT=$2; R=${3:-r3}
F=gpurun_out/final_$TAG; P=gpurun_out/prof_$TAG
cp $P/kernel_stats.csv profiles/${R}_final_lucy1080p_1024spp_kernel_stats.csv
cp $P/pmc_summary.txt profiles/${R}_final_lucy1080p_1024spp_pmc.txt
cp $P/bench_under_rocprof.json profiles/${R}_final_bench_under_rocprof.json
python tools/pmc_to_json.py $P/pmc $COMMIT > profiles/${R}_final_pmc.json
for s in teapot cornell_lucy dual; do
  cp gpurun_out/pmc_$TAG/$s/pmc_summary.txt profiles/${R}_final_${s}1080p_1024spp_pmc.txt
  cp gpurun_out/pmc_$TAG/$s/kernel_stats.csv profiles/${R}_final_${s}1080p_1024spp_kernel_stats.csv
done
cp $F/scene_table.md profiles/${R}_final_scene_table.md
cp $F/example_table.md profiles/${R}_final_example_table.md
cp $F/configs.md profiles/${R}_final_configs.md
cp $F/parity_ids.md profiles/${R}_final_parity.md
grep -v "out of work\|^/opt" $F/stage_split.txt > profiles/${R}_final_stage_split.txt
grep -v "^/opt\|^==" $F/build_probe.txt > profiles/${R}_final_build_probe.txt
python tools/isa_histogram.py > profiles/${R}_final_isa_histogram_now.txt 2>/dev/null
python -c "
import bench, json
p = json.load(open('profiles/${R}_final_pmc.json'))
assert p['kernel_sources_sha'] == bench.kernel_sources_sha(), 'PMC record is not of these kernel sources'
print('PMC record matches the kernel sources')"
