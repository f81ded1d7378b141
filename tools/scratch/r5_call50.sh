#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for a in "--persons 100000 --items 10000 --ability-dim 1" \
         "--persons 1000000 --items 1000 --ability-dim 8 --cond" \
         "--persons 535596 --items 96 --ability-dim 1 --missing 0.2" \
         "--persons 535596 --items 96 --ability-dim 4 --missing 0.2" \
         "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4"; do
  timeout 600 bash tools/ab_libs.sh "$a" h2 cur nt2
done
} > $O/r5_ab50.txt 2>&1
cat $O/r5_ab50.txt
