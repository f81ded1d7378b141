#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for extra in "" "--gather"; do
  rm -rf /tmp/fg; rocprofv3 --pmc FETCH_SIZE -d /tmp/fg -o p -- python $R/tools/profile_kernel.py --iters 3 --persons 1000000 --items 1000 --ability-dim 8 $extra > /tmp/fg.log 2>&1
  echo "== $extra $(grep -o '[0-9.]* ms/call' /tmp/fg.log)"; python $R/tools/rocpd_summary.py /tmp/fg/p_results.db msplit | grep FETCH
done
