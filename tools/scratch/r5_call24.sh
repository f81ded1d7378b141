#!/bin/bash
# round 5, final build: smoke, full GPU suite, bench line + bench profile from ONE box, other paths, shard step, narrow-kernel PMC
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4
timeout 1800 python -m pytest tests -m gpu -q > $O/r5_gpu_tests24.log 2>&1; echo "pytest rc=$?" >> $O/r5_gpu_tests24.log
grep -v '^\.' $O/r5_gpu_tests24.log | grep 'FAILED\|passed\|failed\|rc=' | head -40
timeout 1200 bash tools/collect_profile.sh > /dev/null 2>&1; cp $O/profile_summary.txt $R/profiles/r05_bench_profile.txt; cp $O/profile_summary.txt $O/r5_profile_summary24.txt
timeout 900 python bench.py > $O/r5_bench24.log 2>&1; grep '^{' $O/r5_bench24.log > $O/r5_bench_line24.json; cut -c1-240 $O/r5_bench_line24.json
timeout 1800 bash tools/collect_paths_profile.sh > /dev/null 2>&1; cp $O/paths_profile.txt $O/r5_paths_profile24.txt
timeout 600 bash tools/collect_shard_profile.sh > /dev/null 2>&1; cp $O/shard125k_step_sequence.txt $O/r5_shard24.txt
timeout 600 bash tools/pmc_kernel.sh narrow --persons 535596 --items 96 --ability-dim 1 --missing 0.2 > $O/r5_narrow_pmc24.txt 2>&1
grep 'terms/s' $O/paths_profile.txt | cut -c1-150
