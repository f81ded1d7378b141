#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python tools/scratch/r5_dbg.py > $O/r5_dbg.txt 2>&1; cat $O/r5_dbg.txt
timeout 1200 python -m pytest tests -m gpu -q > $O/r5_gpu_tests2.log 2>&1; echo "pytest rc=$?" >> $O/r5_gpu_tests2.log
grep -v '^\.' $O/r5_gpu_tests2.log | grep 'FAILED\|passed\|failed\|rc=' | head -60
