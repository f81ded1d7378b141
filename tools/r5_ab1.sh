#!/bin/bash
# round 5, GPU call 1: full GPU test suite + same-box A/B of the round-4 library against the tree (panel mode in one launch,
# 3PL guesses in LDS, hook modes)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/r5_gpu_tests1.log 2>&1; echo "pytest rc=$?" >> $O/r5_gpu_tests1.log
tail -3 $O/r5_gpu_tests1.log
S=$O/r5_ab1.txt; : > $S
for a in "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4" \
         "--persons 100000 --items 10000 --ability-dim 1" \
         "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4 --codes" \
         "--persons 1000000 --items 1000 --ability-dim 1 --irt 3" \
         "--persons 1000000 --items 1000 --ability-dim 8 --gather" \
         "--persons 1000000 --items 1000 --ability-dim 8 --flows 4" \
         "--persons 1000000 --items 1000 --ability-dim 8 --given" \
         "--persons 1000000 --items 1000 --ability-dim 8 --cond" \
         "--persons 1000000 --items 1000 --ability-dim 8"; do
  timeout 600 bash tools/ab_libs.sh "$a" r4 cur >> $S 2>&1
done
cat $S
