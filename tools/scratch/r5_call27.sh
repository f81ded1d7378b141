#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for irt in 1 2 3; do for fl in "" "--flows 4"; do for cd in "" "--codes"; do for A in 1 8; do
  a="--persons 100000 --items 10000 --ability-dim $A --irt $irt $fl $cd"
  echo "== $a"
  for k in valu auto; do
    if [ $k = valu ]; then kk="--kernel valu"; else kk=""; fi
    printf "%-5s " $k; python tools/profile_kernel.py $a $kk 2>&1 | tail -1 | sed 's/.*: \([0-9.]* ms\/call\).*ll=\(.*\)/\1  ll=\2/'
  done
done; done; done; done
} > $O/r5_wide27.txt 2>&1
cat $O/r5_wide27.txt
