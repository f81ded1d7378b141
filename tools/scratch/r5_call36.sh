#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/variational-item-response-theory-public_amd/vibo_amd/variants
{
for a in "--persons 1000000 --items 1000 --ability-dim 8 --kernel valu" "--persons 1000000 --items 1000 --ability-dim 1 --kernel valu" "--persons 1000000 --items 1000 --ability-dim 8 --irt 3 --kernel valu" "--persons 1000000 --items 1000 --ability-dim 8 --gather --kernel valu" "--persons 1000000 --items 1000 --ability-dim 8 --codes --kernel valu" "--persons 2048 --items 1000 --ability-dim 8"; do
  echo "== $a"
  for rep in 1 2 3; do for n in cur vsink; do
    if [ "$n" = cur ]; then unset VIBO_HIP_LIB; else export VIBO_HIP_LIB=$V/libvibo_$n.so; fi
    printf "%-6s " $n; python tools/profile_kernel.py $a 2>&1 | tail -1 | sed 's/.*: \([0-9.]* ms\/call\).*ll=\(.*\)/\1  ll=\2/'
  done; done
done
unset VIBO_HIP_LIB
for a in "--persons 535596 --items 96 --ability-dim 1 --missing 0.2 --irt 3" "--persons 535596 --items 96 --ability-dim 1 --missing 0.2 --irt 3 --codes"; do
  timeout 600 bash tools/ab_libs.sh "$a" sink cur
done
} > $O/r5_ab36.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_narrow.py -m gpu -q -x 2>&1 | tail -2
cat $O/r5_ab36.txt
