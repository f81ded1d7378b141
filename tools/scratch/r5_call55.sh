#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for a in "--persons 1000000 --items 1000 --ability-dim 8" \
         "--persons 1000000 --items 1000 --ability-dim 8 --codes" \
         "--persons 1000000 --items 1000 --ability-dim 8 --irt 3" \
         "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4"; do
  timeout 600 bash tools/ab_libs.sh "$a" cur mit
done
timeout 600 bash tools/ab_libs.sh "--persons 1000000 --items 1000 --ability-dim 1 --cond" cur cit
} > $O/r5_ab55.txt 2>&1
cat $O/r5_ab55.txt
