#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/variational-item-response-theory-public_amd/vibo_amd/variants
timeout 1800 python -m pytest tests -m gpu -q -x > $O/r5_gpu_tests40.log 2>&1; echo "pytest rc=$?" >> $O/r5_gpu_tests40.log
tail -3 $O/r5_gpu_tests40.log; grep -n "^E \|FAILED" $O/r5_gpu_tests40.log | head
{
for m in deep link; do
  echo "== decoder $m"
  for rep in 1 2 3; do for n in h1 cur; do
    if [ "$n" = cur ]; then unset VIBO_HIP_LIB; else export VIBO_HIP_LIB=$V/libvibo_$n.so; fi
    printf "%-6s " $n; python tools/profile_decoder.py --mode $m --iters 5 2>&1 | tail -1 | cut -c1-120
  done; done
done
unset VIBO_HIP_LIB
for a in "--persons 1000000 --items 1000 --ability-dim 8 --flows 4" \
         "--persons 1000000 --items 1000 --ability-dim 1 --irt 3 --flows 4" \
         "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4"; do
  timeout 600 bash tools/ab_libs.sh "$a" h1 cur
done
} > $O/r5_ab40.txt 2>&1
cat $O/r5_ab40.txt
