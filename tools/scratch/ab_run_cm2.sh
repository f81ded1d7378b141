#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=$R/variational-item-response-theory-public_amd/vibo_amd/variants
for n in "$@"; do
  for A in 1 8; do
  rm -rf /tmp/pc
  VIBO_HIP_LIB=$V/libvibo_$n.so timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/pc -o kt -- python $R/tools/profile_kernel.py --iters 3 --persons 1000000 --items 1000 --ability-dim $A --cond --codes > /tmp/pc.log 2>&1
  echo "== $n A=$A $(grep -o 'll=.*' /tmp/pc.log)"; python $R/tools/rocpd_summary.py /tmp/pc/kt_results.db cm_ | grep -E "forward|backward_kernel" | cut -c1-120
  done
done
