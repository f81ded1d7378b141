#!/bin/bash
# round 5: the 3PL tile as v = n / t (one exponential, one reciprocal per cell) -- full GPU suite + same-box A/B against the committed build (c5)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/r5_gpu_tests6.log 2>&1; echo "pytest rc=$?" >> $O/r5_gpu_tests6.log
grep -v '^\.' $O/r5_gpu_tests6.log | grep 'FAILED\|passed\|failed\|rc=' | head -40
S=$O/r5_ab6.txt; : > $S
for a in "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4" \
         "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4 --codes" \
         "--persons 1000000 --items 1000 --ability-dim 1 --irt 3" \
         "--persons 1000000 --items 1000 --ability-dim 8 --irt 3" \
         "--persons 1000000 --items 1000 --ability-dim 8 --irt 3 --gather" \
         "--persons 1000000 --items 1000 --ability-dim 8 --irt 3 --codes" \
         "--persons 1000000 --items 1000 --ability-dim 8 --irt 3 --no-grad" \
         "--persons 1000000 --items 1000 --ability-dim 8"; do
  timeout 600 bash tools/ab_libs.sh "$a" c5 cur >> $S 2>&1
done
cat $S
