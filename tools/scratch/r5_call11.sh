#!/bin/bash
# round 5: 3PL check on the extrema; row_cnt prefetch reverted; where does the wide plain call lose 0.2 ms against the committed build?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/variational-item-response-theory-public_amd/vibo_amd/variants
S=$O/r5_ab11.txt; : > $S
for a in "--persons 100000 --items 10000 --ability-dim 1" \
         "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4" \
         "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4 --codes" \
         "--persons 1000000 --items 1000 --ability-dim 8 --irt 3" \
         "--persons 1000000 --items 1000 --ability-dim 8 --irt 3 --flows 4" \
         "--persons 1000000 --items 1000 --ability-dim 1 --irt 3 --item-scale 4" \
         "--persons 1000000 --items 1000 --ability-dim 1 --irt 3"; do
  timeout 600 bash tools/ab_libs.sh "$a" c5 cur >> $S 2>&1
done
cat $S
cd /tmp && export TMPDIR=/tmp
for n in c5 cur; do
  if [ $n = cur ]; then unset VIBO_HIP_LIB; else export VIBO_HIP_LIB=$V/libvibo_$n.so; fi
  rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/tools/profile_kernel.py --iters 3 --persons 100000 --items 10000 --ability-dim 1 > /tmp/kt.log 2>&1
  echo "== $n"; grep "terms/s" /tmp/kt.log; python $R/tools/rocpd_summary.py /tmp/kt/kt_results.db vibo | cut -c1-150
done > $O/r5_wide11.txt 2>&1
unset VIBO_HIP_LIB
cat $O/r5_wide11.txt
cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/r5_gpu_tests11.log 2>&1; echo "pytest rc=$?" >> $O/r5_gpu_tests11.log
grep -v '^\.' $O/r5_gpu_tests11.log | grep 'FAILED\|passed\|failed\|rc=' | head -40
