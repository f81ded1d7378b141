#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export VIBO_HIP_LIB=$R/variational-item-response-theory-public_amd/vibo_amd/variants/libvibo_timing.so
{
for a in "--persons 125000 --items 1000 --ability-dim 8" "--persons 100000 --items 1000 --ability-dim 1" "--persons 1000000 --items 1000 --ability-dim 8"; do
  echo "== $a"; timeout 300 python tools/ms_timing.py $a 2>&1 | tail -21
done
} > $O/r5_timing43.txt 2>&1
cat $O/r5_timing43.txt
