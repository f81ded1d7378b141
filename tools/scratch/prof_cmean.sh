#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
python $R/tools/profile_cmean.py 2>&1 | tail -2
rm -rf /tmp/pcm; rocprofv3 --kernel-trace --stats -d /tmp/pcm -o kt -- python $R/tools/profile_cmean.py --no-gemm > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pcm/kt_results.db cm_ | cut -c1-150
rm -rf /tmp/pcm2; rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE -d /tmp/pcm2 -o p -- python $R/tools/profile_cmean.py --no-gemm > /dev/null 2>&1
python $R/tools/rocpd_summary.py /tmp/pcm2/p_results.db cm_ | grep -E "forward|backward_k|backward_wide" | cut -c1-150
