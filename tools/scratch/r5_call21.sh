#!/bin/bash
# round 5: fuzz campaign on the current build (the elbo target now also pins the matrix / VALU kernels per case)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for seed in 501 502 503; do echo "== elbo seed $seed"; timeout 400 python tools/fuzz_parity.py --seconds 240 --seed $seed 2>&1 | grep -v amdgpu.ids | tail -4; done
echo "== module"; timeout 300 python tools/fuzz_parity.py --target module --seconds 150 --seed 504 2>&1 | grep -v amdgpu.ids | tail -3
echo "== trainer"; timeout 300 python tools/fuzz_parity.py --target trainer --seconds 150 --seed 505 2>&1 | grep -v amdgpu.ids | tail -3
echo "== multi"; timeout 200 python tools/fuzz_parity.py --target multi --seconds 90 --seed 506 2>&1 | grep -v amdgpu.ids | tail -3
} > $O/r5_fuzz21.txt 2>&1
cat $O/r5_fuzz21.txt | cut -c1-400
