#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/variational-item-response-theory-public_amd/vibo_amd/variants
timeout 900 python -m pytest tests/test_gpu_decoder.py -m gpu -q -x 2>&1 | tail -2
{
for m in deep link; do
  echo "== decoder $m"
  for rep in 1 2; do for n in h3 cur; do
    if [ "$n" = cur ]; then unset VIBO_HIP_LIB; else export VIBO_HIP_LIB=$V/libvibo_$n.so; fi
    printf "%-6s " $n; python tools/profile_decoder.py --mode $m --iters 5 2>&1 | tail -1 | cut -c1-100
  done; done
done
unset VIBO_HIP_LIB
timeout 200 python tools/fuzz_decoder.py --seconds 60 --seed 78 2>&1 | tail -1
} > $O/r5_ab59.txt 2>&1
cat $O/r5_ab59.txt
