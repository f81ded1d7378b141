#!/bin/bash
# round 5: what the in-launch count exchange costs at ten panels -- poll interval, no polling (wrong results, timing only), no store wait
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
S=$O/r5_ab19.txt; : > $S
for a in "--persons 100000 --items 10000 --ability-dim 1" "--persons 100000 --items 10000 --ability-dim 1 --codes" "--persons 200000 --items 5000 --ability-dim 1" "--persons 100000 --items 3000 --ability-dim 1"; do
  timeout 600 bash tools/ab_libs.sh "$a" c5 cur xs32 xs127 xnopoll xnowait >> $S 2>&1
done
cat $S
