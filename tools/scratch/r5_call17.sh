#!/bin/bash
# round 5: wide plain rows WITHOUT the count pass (counts exchanged by the panels' workgroups inside the launch)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "general_kernel_vs_oracle or ragged or forward_only" > $O/r5_wide17.log 2>&1; echo "pytest rc=$?" >> $O/r5_wide17.log
tail -12 $O/r5_wide17.log | cut -c1-250
S=$O/r5_ab17.txt; : > $S
for a in "--persons 100000 --items 10000 --ability-dim 1" "--persons 100000 --items 10000 --ability-dim 8" "--persons 100000 --items 10000 --ability-dim 1 --codes" "--persons 200000 --items 5000 --ability-dim 8 --irt 3" "--persons 20000 --items 2500 --ability-dim 4 --flows 2" "--persons 100000 --items 10000 --ability-dim 1 --gather"; do
  timeout 300 bash tools/ab_libs.sh "$a" c5 cur >> $S 2>&1
done
cat $S
timeout 1800 python -m pytest tests -m gpu -q > $O/r5_gpu_tests17.log 2>&1; echo "pytest rc=$?" >> $O/r5_gpu_tests17.log
grep -v '^\.' $O/r5_gpu_tests17.log | grep 'FAILED\|passed\|failed\|rc=' | head -40
