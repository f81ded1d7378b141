#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pct; rocprofv3 --kernel-trace --stats -d /tmp/pct -o kt -- python $R/tools/step_time.py --irt 3 --items 1000 --batch 16 --cond --flows 4 > /tmp/pct.log 2>&1
python $R/tools/rocpd_summary.py /tmp/pct/kt_results.db | grep -i "vibo\|ct_\|kernel  " | head -30
