#!/bin/bash
# round 5 profiles of the current build: bench line, bench profile (kernel trace + PMC + config 2), other paths, 125k shard step, narrow kernel PMC
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python bench.py > $O/r5_bench12.log 2>&1; grep '^{' $O/r5_bench12.log > $O/r5_bench_line12.json; cut -c1-300 $O/r5_bench_line12.json
timeout 1200 bash tools/collect_profile.sh > /dev/null 2>&1; cp $O/profile_summary.txt $O/r5_profile_summary12.txt
timeout 1800 bash tools/collect_paths_profile.sh > /dev/null 2>&1; cp $O/paths_profile.txt $O/r5_paths_profile12.txt
timeout 600 bash tools/collect_shard_profile.sh > /dev/null 2>&1; cp $O/shard125k_step_sequence.txt $O/r5_shard12.txt
timeout 600 bash tools/pmc_kernel.sh narrow --persons 535596 --items 96 --ability-dim 1 --missing 0.2 > $O/r5_narrow_pmc12.txt 2>&1
grep 'terms/s' $O/paths_profile.txt | cut -c1-150; tail -12 $O/r5_shard12.txt | cut -c1-160; head -30 $O/r5_narrow_pmc12.txt | cut -c1-160
