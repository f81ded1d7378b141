#!/bin/bash
# round 5: 3PL tile in halves (spill-free) -- full GPU suite (with the new at-size tests) + same-box A/B against the committed build (c5); narrow rows on the matrix kernel
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/r5_gpu_tests7.log 2>&1; echo "pytest rc=$?" >> $O/r5_gpu_tests7.log
grep -v '^\.' $O/r5_gpu_tests7.log | grep 'FAILED\|passed\|failed\|rc=' | head -40
S=$O/r5_ab7.txt; : > $S
for a in "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4" \
         "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4 --codes" \
         "--persons 1000000 --items 1000 --ability-dim 1 --irt 3" \
         "--persons 1000000 --items 1000 --ability-dim 8 --irt 3" \
         "--persons 1000000 --items 1000 --ability-dim 8 --irt 3 --gather" \
         "--persons 1000000 --items 1000 --ability-dim 8 --irt 3 --codes" \
         "--persons 1000000 --items 1000 --ability-dim 8 --irt 3 --flows 4" \
         "--persons 1000000 --items 1000 --ability-dim 8 --irt 3 --no-grad"; do
  timeout 600 bash tools/ab_libs.sh "$a" c5 cur >> $S 2>&1
done
cat $S
for k in auto matrix valu; do for sh in "--persons 535596 --items 96 --ability-dim 1 --missing 0.2" "--persons 535596 --items 128 --ability-dim 1 --missing 0.2" "--persons 8000 --items 100 --ability-dim 1"; do
  echo "== $sh --kernel $k"; python tools/profile_kernel.py $sh --kernel $k 2>&1 | tail -1; done; done
