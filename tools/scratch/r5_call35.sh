#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for a in "--persons 125000 --items 1000 --ability-dim 8" "--persons 100000 --items 1000 --ability-dim 1" "--persons 125000 --items 1000 --ability-dim 8"; do
  timeout 600 bash tools/ab_libs.sh "$a" r5f cur sink
done
} > $O/r5_ab35.txt 2>&1
cat $O/r5_ab35.txt
