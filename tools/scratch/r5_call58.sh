#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/variational-item-response-theory-public_amd/vibo_amd/variants
{
for m in deep link; do
  echo "== decoder $m"
  for rep in 1 2; do for n in cur dvf dvf2; do
    if [ "$n" = cur ]; then unset VIBO_HIP_LIB; else export VIBO_HIP_LIB=$V/libvibo_$n.so; fi
    printf "%-6s " $n; python tools/profile_decoder.py --mode $m --iters 5 2>&1 | tail -1 | cut -c1-100
  done; done
done
} > $O/r5_ab58.txt 2>&1
cat $O/r5_ab58.txt
