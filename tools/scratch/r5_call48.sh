#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for a in "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4" \
         "--persons 1000000 --items 1000 --ability-dim 1 --cond"; do
  timeout 600 bash tools/ab_libs.sh "$a" h2 nts ntl
done
} > $O/r5_ab48.txt 2>&1
cat $O/r5_ab48.txt
