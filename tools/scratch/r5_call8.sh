#!/bin/bash
# round 5: 3PL second path = the same formula clamped; cond_pre with two row batches in flight; the narrow-row kernel
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_narrow.py -m gpu -q -x > $O/r5_gpu_narrow8.log 2>&1; echo "pytest rc=$?" >> $O/r5_gpu_narrow8.log
grep -v '^\.' $O/r5_gpu_narrow8.log | grep 'FAILED\|passed\|failed\|rc=\|Error' | head -20
for sh in "--persons 535596 --items 96 --ability-dim 1 --missing 0.2" "--persons 535596 --items 128 --ability-dim 1 --missing 0.2" "--persons 8000 --items 100 --ability-dim 1" "--persons 535596 --items 96 --ability-dim 1 --missing 0.2 --codes" "--persons 535596 --items 96 --ability-dim 4 --missing 0.2" "--persons 535596 --items 64 --ability-dim 1" "--persons 535596 --items 96 --ability-dim 1 --irt 3"; do
  for k in auto valu; do echo "== $sh --kernel $k"; timeout 300 python tools/profile_kernel.py $sh --kernel $k 2>&1 | tail -1; done; done > $O/r5_narrow8.txt 2>&1
cat $O/r5_narrow8.txt
timeout 1800 python -m pytest tests -m gpu -q > $O/r5_gpu_tests8.log 2>&1; echo "pytest rc=$?" >> $O/r5_gpu_tests8.log
grep -v '^\.' $O/r5_gpu_tests8.log | grep 'FAILED\|passed\|failed\|rc=' | head -40
S=$O/r5_ab8.txt; : > $S
for a in "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4" \
         "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4 --codes" \
         "--persons 1000000 --items 1000 --ability-dim 8 --irt 3" \
         "--persons 1000000 --items 1000 --ability-dim 8 --irt 3 --flows 4" \
         "--persons 1000000 --items 1000 --ability-dim 1 --cond" \
         "--persons 1000000 --items 1000 --ability-dim 4 --cond"; do
  timeout 600 bash tools/ab_libs.sh "$a" c5 cur >> $S 2>&1
done
cat $S
