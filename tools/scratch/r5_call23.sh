#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "golden_adam_trajectory_through_the_fused_trainers" > $O/r5_traj23.log 2>&1; echo "pytest rc=$?" >> $O/r5_traj23.log
grep -v "^\." $O/r5_traj23.log | tail -30 | cut -c1-300
