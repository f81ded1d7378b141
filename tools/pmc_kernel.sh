#!/bin/bash
# PMC counters for one kernel configuration of tools/profile_kernel.py (run on the GPU box).
#   tools/pmc_kernel.sh <name-filter> <profile_kernel.py args...>     -> text on stdout
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
F=$1; shift
W=/tmp/vibo_pmc; rm -rf $W; mkdir -p $W
cd /tmp && export TMPDIR=/tmp
B="python $R/tools/${PROFILE_SCRIPT:-profile_kernel.py} --iters 5 $*"
echo "# $B"
rocprofv3 --kernel-trace --stats -d $W/kt -o kt -- $B > $W/kt.log 2>&1
python $R/tools/rocpd_summary.py $W/kt/kt_results.db $F | head -6
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; do
  n=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass -d $W/$n -o p -- $B > $W/$n.log 2>&1 || tail -3 $W/$n.log
  python $R/tools/rocpd_summary.py $W/$n/p_results.db $F | grep -A40 "^PMC" | grep -v "^PMC"
done
rm -rf $W
