#!/bin/bash
# Same-box A/B of the variants built by tools/ab_build.sh: ab_run.sh "<profile_kernel.py args>" name [name ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=$R/variational-item-response-theory-public_amd/vibo_amd/variants
A=$1; shift
printf "%-6s " ref; python $R/tools/profile_kernel.py $A --kernel valu 2>&1 | tail -1 | sed "s/.*: \([0-9.]* ms\/call\).*/\1/"
for rep in 1 2 3; do
  for n in "$@"; do
    printf "%-6s " $n; VIBO_HIP_LIB=$V/libvibo_$n.so python $R/tools/profile_kernel.py $A 2>&1 | tail -1 | sed 's/.*: \([0-9.]* ms\/call\).*/\1/'
  done
done
