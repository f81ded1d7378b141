#!/bin/bash
# Build kernel variants side by side for same-box A/B timing (tools/ab_run.sh): ab_build.sh name "-DFLAG ..." [name flags ...]
# -> variational-item-response-theory-public_amd/vibo_amd/variants/libvibo_<name>.so (select with VIBO_HIP_LIB); each variant has
# its own object directory (csrc/build_<name>), the in-tree build is left alone.  name = "timing": make TIMING=1 (tools/ms_timing.py)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/variational-item-response-theory-public_amd/csrc
V=$R/variational-item-response-theory-public_amd/vibo_amd/variants
mkdir -p $V
while [ $# -ge 2 ]; do
  n=$1; fl=$2; shift 2
  T=""; [ "$n" = timing ] && T="TIMING=1"
  make -C $C -j8 $T XDEF="$fl" B=build_$n OUT=$V/libvibo_$n.so 2>&1 | grep -E "error|Error" || true
  ls -la $V/libvibo_$n.so
done
