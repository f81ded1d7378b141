#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
timeout 400 python tools/fuzz_parity.py --seconds 240 --seed 9051 --target elbo 2>&1 | grep -v amdgpu.ids | tail -6
timeout 200 python tools/fuzz_parity.py --seconds 90 --seed 9052 --target module 2>&1 | grep -v amdgpu.ids | tail -4
timeout 200 python tools/fuzz_parity.py --seconds 90 --seed 9053 --target multi 2>&1 | grep -v amdgpu.ids | tail -4
timeout 300 python tools/fuzz_parity.py --seconds 150 --seed 9054 --target trainer 2>&1 | grep -v amdgpu.ids | tail -6
timeout 200 python tools/fuzz_decoder.py --seconds 90 --seed 9055 2>&1 | grep -v amdgpu.ids | tail -4
} > $O/r5_fuzz45.txt 2>&1
cat $O/r5_fuzz45.txt
