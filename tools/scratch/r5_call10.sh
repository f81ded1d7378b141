#!/bin/bash
# round 5: 3PL saturation flag per batch, row counts a batch ahead (unconditional load), narrow kernel with 4-row units
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/r5_gpu_tests10.log 2>&1; echo "pytest rc=$?" >> $O/r5_gpu_tests10.log
grep -v '^\.' $O/r5_gpu_tests10.log | grep 'FAILED\|passed\|failed\|rc=' | head -40
S=$O/r5_ab10.txt; : > $S
for a in "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4" \
         "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4 --codes" \
         "--persons 1000000 --items 1000 --ability-dim 8 --irt 3" \
         "--persons 1000000 --items 1000 --ability-dim 8 --irt 3 --flows 4" \
         "--persons 1000000 --items 1000 --ability-dim 1 --irt 3 --item-scale 4" \
         "--persons 100000 --items 10000 --ability-dim 1" \
         "--persons 100000 --items 10000 --ability-dim 8"; do
  timeout 600 bash tools/ab_libs.sh "$a" c5 cur >> $S 2>&1
done
for a in "--persons 535596 --items 96 --ability-dim 1 --missing 0.2" "--persons 535596 --items 96 --ability-dim 2 --missing 0.2" "--persons 535596 --items 96 --ability-dim 4 --missing 0.2" "--persons 535596 --items 64 --ability-dim 1" "--persons 535596 --items 96 --ability-dim 1 --irt 3"; do
  timeout 600 bash tools/ab_libs.sh "$a" nwp1 cur >> $S 2>&1
done
cat $S
