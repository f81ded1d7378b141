#!/bin/bash
# Same-box A/B of library builds on the folded train step: ab_insitu.sh "<shapes>" lib [lib ...]   (lib = cur | a name under
# vibo_amd/variants/); three interleaved rounds of tools/ab_insitu.py.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=$R/variational-item-response-theory-public_amd/vibo_amd/variants
S=$1; shift
for rep in 1 2 3; do
  for n in "$@"; do
    if [ "$n" = cur ]; then unset VIBO_HIP_LIB; T=cur; else export VIBO_HIP_LIB=$V/libvibo_$n.so; T=$n; fi
    python $R/tools/ab_insitu.py $S --tag $T 2>&1 | grep -v amdgpu.ids
  done
done
unset VIBO_HIP_LIB
