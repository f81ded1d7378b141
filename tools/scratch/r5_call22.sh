#!/bin/bash
# round 5: the trainer fuzz report of seed 505 on the committed round-4-end library (c5): the same case fails there?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
V=$R/variational-item-response-theory-public_amd/vibo_amd/variants
{
echo "== trainer seed 505, library c5 (commit 87f0f3d)"; VIBO_HIP_LIB=$V/libvibo_c5.so timeout 400 python tools/fuzz_parity.py --target trainer --seconds 200 --seed 505 2>&1 | grep -v amdgpu.ids | tail -3
echo "== trainer seed 507, current library"; timeout 400 python tools/fuzz_parity.py --target trainer --seconds 200 --seed 507 2>&1 | grep -v amdgpu.ids | tail -3
} > $O/r5_fuzz22.txt 2>&1
cat $O/r5_fuzz22.txt | cut -c1-300
