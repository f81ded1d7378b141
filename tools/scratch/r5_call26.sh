#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for a in "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4" \
         "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4 --codes" \
         "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --flows 4" \
         "--persons 100000 --items 10000 --ability-dim 1 --flows 4" \
         "--persons 1000000 --items 1000 --ability-dim 1 --irt 3 --flows 4" \
         "--persons 1000000 --items 1000 --ability-dim 8 --flows 4" \
         "--persons 1000000 --items 1000 --ability-dim 8 --irt 3 --flows 4 --gather"; do
  timeout 600 bash tools/ab_libs.sh "$a" r5f cur
done
} > $O/r5_ab26.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x -k "flow or config5 or cond" > $O/r5_gpu_tests26.log 2>&1; echo "pytest rc=$?" >> $O/r5_gpu_tests26.log
tail -3 $O/r5_gpu_tests26.log
cat $O/r5_ab26.txt
