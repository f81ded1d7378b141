#!/bin/bash
# conditional posterior: matrix-pipe passes vs the VALU ones, same box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {
  rm -rf /tmp/pc; rocprofv3 --kernel-trace --stats -d /tmp/pc -o kt -- python $R/tools/profile_kernel.py --iters 3 "$@" 2>&1 | grep "ms/call"
  python $R/tools/rocpd_summary.py /tmp/pc/kt_results.db vibo | cut -c1-150
}
for c in "$@"; do echo "== $c"; run $c; done
