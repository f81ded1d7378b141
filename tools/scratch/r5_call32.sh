#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for a in "--persons 1000000 --items 1000 --ability-dim 8 --irt 3" \
         "--persons 1000000 --items 1000 --ability-dim 1 --irt 3" \
         "--persons 1000000 --items 1000 --ability-dim 8 --irt 3 --codes" \
         "--persons 1000000 --items 1000 --ability-dim 8 --irt 3 --gather" \
         "--persons 1000000 --items 1000 --ability-dim 8 --irt 3 --flows 4" \
         "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4"; do
  timeout 600 bash tools/ab_libs.sh "$a" sink snp
done
for a in "--persons 100000 --items 10000 --ability-dim 1" \
         "--persons 100000 --items 10000 --ability-dim 8" \
         "--persons 100000 --items 10000 --ability-dim 1 --codes" \
         "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --flows 4" \
         "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4" \
         "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4 --codes"; do
  timeout 600 bash tools/ab_libs.sh "$a" sink prc
done
} > $O/r5_ab32.txt 2>&1
cat $O/r5_ab32.txt
