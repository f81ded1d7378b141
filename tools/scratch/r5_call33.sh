#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q -x > $O/r5_gpu_tests33.log 2>&1; echo "pytest rc=$?" >> $O/r5_gpu_tests33.log
tail -3 $O/r5_gpu_tests33.log
{
for a in "--persons 1000000 --items 1000 --ability-dim 8" \
         "--persons 1000000 --items 1000 --ability-dim 8 --irt 3" \
         "--persons 1000000 --items 1000 --ability-dim 8 --irt 3 --gather" \
         "--persons 1000000 --items 1000 --ability-dim 8 --gather" \
         "--persons 100000 --items 10000 --ability-dim 1" \
         "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4" \
         "--persons 1000000 --items 1000 --ability-dim 8 --cond" \
         "--persons 125000 --items 1000 --ability-dim 8"; do
  timeout 600 bash tools/ab_libs.sh "$a" r5f cur
done
} > $O/r5_ab33.txt 2>&1
cat $O/r5_ab33.txt
