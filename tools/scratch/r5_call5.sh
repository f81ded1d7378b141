#!/bin/bash
# round 5, re-entry GPU call: full GPU suite, default bench line, bench profile + other-paths trace of the committed tree
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/r5_gpu_tests5.log 2>&1; echo "pytest rc=$?" >> $O/r5_gpu_tests5.log
grep -v '^\.' $O/r5_gpu_tests5.log | grep 'FAILED\|passed\|failed\|rc=' | head -40
timeout 900 python bench.py > $O/r5_bench5.log 2>&1; grep '^{' $O/r5_bench5.log | cut -c1-600
timeout 900 bash tools/collect_profile.sh > /dev/null 2>&1; cp $O/profile_summary.txt $O/r5_profile_summary5.txt
timeout 1500 bash tools/collect_paths_profile.sh > /dev/null 2>&1; cp $O/paths_profile.txt $O/r5_paths_profile5.txt
grep 'terms/s' $O/paths_profile.txt | cut -c1-160
