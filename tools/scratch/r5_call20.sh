#!/bin/bash
# round 5: person-sharded FusedMeanTrainer (two gloo ranks on one device) + the distributed / trainer / mean-merge suites
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_mean_merge.py tests/test_gpu_trainer.py -m gpu -q > $O/r5_dist20.log 2>&1; echo "pytest rc=$?" >> $O/r5_dist20.log
tail -30 $O/r5_dist20.log | cut -c1-300
