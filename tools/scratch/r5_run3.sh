#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
V=$R/variational-item-response-theory-public_amd/vibo_amd/variants
for n in cur mixnop; do
  echo "=== $n"
  if [ $n = cur ]; then unset VIBO_HIP_LIB; else export VIBO_HIP_LIB=$V/libvibo_$n.so; fi
  python tools/scratch/r5_dbg.py 2>&1 | grep -v amdgpu.ids | grep '^(\|bad entries'
done
