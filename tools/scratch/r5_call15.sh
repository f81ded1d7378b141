#!/bin/bash
# round 5: narrow kernel v2 (per-person math once per 64-row unit) -- tests + A/B against v1 (variants/libvibo_nwp1.so)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_narrow.py -m gpu -q -x > $O/r5_narrow15.log 2>&1; echo "pytest rc=$?" >> $O/r5_narrow15.log
tail -15 $O/r5_narrow15.log | cut -c1-250
S=$O/r5_ab15.txt; : > $S
for a in "--persons 535596 --items 96 --ability-dim 1 --missing 0.2" "--persons 535596 --items 96 --ability-dim 2 --missing 0.2" "--persons 535596 --items 96 --ability-dim 4 --missing 0.2" "--persons 535596 --items 64 --ability-dim 1" "--persons 535596 --items 96 --ability-dim 1 --irt 3" "--persons 535596 --items 96 --ability-dim 1 --missing 0.2 --codes" "--persons 8000 --items 100 --ability-dim 1" "--persons 16 --items 100 --ability-dim 1"; do
  timeout 600 bash tools/ab_libs.sh "$a" nwp1 cur >> $S 2>&1
done
cat $S
