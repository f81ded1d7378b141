#!/bin/bash
# round 5: cond_pre as a wave-per-row stream (one ability dim)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/r5_gpu_tests14.log 2>&1; echo "pytest rc=$?" >> $O/r5_gpu_tests14.log
grep -v '^\.' $O/r5_gpu_tests14.log | grep 'FAILED\|passed\|failed\|rc=' | head -40
S=$O/r5_ab14.txt; : > $S
for a in "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4" \
         "--persons 1000000 --items 1000 --ability-dim 1 --cond" \
         "--persons 1000000 --items 1000 --ability-dim 1 --cond --codes" \
         "--persons 1000000 --items 1000 --ability-dim 1 --cond --gather" \
         "--persons 65536 --items 1000 --ability-dim 1 --cond"; do
  timeout 600 bash tools/ab_libs.sh "$a" c5 cur >> $S 2>&1
done
cat $S
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/tools/profile_kernel.py --iters 3 --persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4 > /tmp/kt.log 2>&1
grep "terms/s" /tmp/kt.log; python $R/tools/rocpd_summary.py /tmp/kt/kt_results.db vibo | cut -c1-150
