#!/bin/bash
# Same-box A/B of whole library builds: ab_libs.sh "<profile_kernel.py args>" lib [lib ...]
# lib = "cur" (the in-tree build) or a name under vibo_amd/variants/ (libvibo_<name>.so); three interleaved rounds.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=$R/variational-item-response-theory-public_amd/vibo_amd/variants
A=$1; shift
echo "== $A"
# per-box calibration: the VALU row-split kernel (round 1 code, untouched) of the in-tree build on the same arguments
printf "%-6s " valu; python $R/tools/profile_kernel.py $A --kernel valu 2>&1 | tail -1 | sed 's/.*: \([0-9.]* ms\/call\).*ll=\(.*\)/\1  ll=\2/'
for rep in 1 2 3; do
  for n in "$@"; do
    if [ "$n" = cur ]; then unset VIBO_HIP_LIB; else export VIBO_HIP_LIB=$V/libvibo_$n.so; fi
    printf "%-6s " $n; python $R/tools/profile_kernel.py $A 2>&1 | tail -1 | sed 's/.*: \([0-9.]* ms\/call\).*ll=\(.*\)/\1  ll=\2/'
  done
done
unset VIBO_HIP_LIB
