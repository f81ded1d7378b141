#!/bin/bash
# variants of vibo_cmean.hip side by side: ab_cmean.sh name "-DFLAG ..." [name flags ...] -> vibo_amd/variants/libvibo_<name>.so
set -e
R=/root/repo
C=$R/variational-item-response-theory-public_amd/csrc
V=$R/variational-item-response-theory-public_amd/vibo_amd/variants
mkdir -p $V
OTHERS=$(ls $C/build/*.o | grep -v "vibo_cmean")
while [ $# -ge 2 ]; do
  n=$1; fl=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $fl -c $C/vibo_cmean.hip -o /tmp/vibo_cmean_$n.o 2>&1 | grep -E "error" || true
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libvibo_$n.so $OTHERS /tmp/vibo_cmean_$n.o
  ls -la $V/libvibo_$n.so
done
