#!/bin/bash
# memory-side counters for one profile_kernel.py configuration:  tools/scratch/pmc_mem.sh <name-filter> <args...>
R=${GRAFT_REPO_ROOT:-/root/repo}
F=$1; shift
W=/tmp/vibo_pmc2; rm -rf $W; mkdir -p $W
cd /tmp && export TMPDIR=/tmp
B="python $R/tools/profile_kernel.py --iters 3 $*"
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCC_TAG_STALL_sum TCC_EA0_RDREQ_32B_sum TCC_BUSY_sum"; do
  n=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass -d $W/$n -o p -- $B > $W/$n.log 2>&1 || { echo "pass failed: $pass"; tail -2 $W/$n.log; }
  python $R/tools/rocpd_summary.py $W/$n/p_results.db $F 2>/dev/null | grep -A60 "^PMC" | grep -v "^PMC"
done
rm -rf $W
