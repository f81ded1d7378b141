#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export VIBO_HIP_LIB=$R/variational-item-response-theory-public_amd/vibo_amd/variants/libvibo_timing.so
{
for a in "--persons 1000000 --items 1000 --ability-dim 8" \
         "--persons 100000 --items 10000 --ability-dim 1" \
         "--persons 100000 --items 10000 --ability-dim 1 --codes" \
         "--persons 1000000 --items 1000 --ability-dim 1 --cond --codes"; do
  echo "== $a"; timeout 300 python tools/ms_timing.py $a 2>&1 | tail -22
done
} > $O/r5_timing30.txt 2>&1
cat $O/r5_timing30.txt
