#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/variational-item-response-theory-public_amd/vibo_amd/variants
{
echo "== r5f"; VIBO_HIP_LIB=$V/libvibo_r5f.so timeout 400 python tools/fuzz_parity.py --seconds 200 --seed 9054 --target trainer 2>&1 | grep -v amdgpu.ids | tail -3
echo "== cur"; timeout 400 python tools/fuzz_parity.py --seconds 200 --seed 9054 --target trainer 2>&1 | grep -v amdgpu.ids | tail -3
} > $O/r5_fuzz46.txt 2>&1
cat $O/r5_fuzz46.txt
