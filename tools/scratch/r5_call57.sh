#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for a in "--persons 1000000 --items 1000 --ability-dim 8 --cond" \
         "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4"; do
  timeout 600 bash tools/ab_libs.sh "$a" cur cmx
done
for a in "--persons 1000000 --items 1000 --ability-dim 8 --flows 4" \
         "--persons 1000000 --items 1000 --ability-dim 8 --gather" \
         "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --flows 4"; do
  timeout 600 bash tools/ab_libs.sh "$a" cur mit2
done
} > $O/r5_ab57.txt 2>&1
cat $O/r5_ab57.txt
