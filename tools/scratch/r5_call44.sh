#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for a in "--persons 125000 --items 1000 --ability-dim 8" \
         "--persons 100000 --items 1000 --ability-dim 1" \
         "--persons 100000 --items 1000 --ability-dim 1 --codes" \
         "--persons 125000 --items 1000 --ability-dim 8" \
         "--persons 1000000 --items 1000 --ability-dim 8" \
         "--persons 100000 --items 10000 --ability-dim 1" \
         "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4"; do
  timeout 600 bash tools/ab_libs.sh "$a" h2 cur
done
} > $O/r5_ab44.txt 2>&1
timeout 1800 python -m pytest tests -m gpu -q -x > $O/r5_gpu_tests44.log 2>&1; echo "pytest rc=$?" >> $O/r5_gpu_tests44.log
tail -3 $O/r5_gpu_tests44.log; grep -n "^E \|FAILED" $O/r5_gpu_tests44.log | head
cat $O/r5_ab44.txt
