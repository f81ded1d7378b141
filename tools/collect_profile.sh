#!/bin/bash
# Run on the GPU box: rocprofv3 evidence for `python bench.py --steps 5 --warmup 2 --no-cpu-baseline`.
# Writes a text summary to gpurun_out/profile_summary.txt (the rocpd databases are deleted: too large to copy back).
# Counter passes follow MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE in separate --pmc passes, no tracing flags
# other than the implicit kernel dispatch table.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
W=/tmp/vibo_prof; rm -rf $W; mkdir -p $W
cd /tmp && export TMPDIR=/tmp
# (--also-ability-dim 0 --no-also-config2: those legs run the same kernel instantiation and would blur the per-kernel averages;
#  config 2 -- 100k x 1k, ability_dim 1 -- gets its own kernel trace at the end)
B="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --also-ability-dim 0 --no-also-config2 $BENCH_ARGS"
S=$OUT/profile_summary.txt
{
echo "# command: rocprofv3 --kernel-trace --stats -- $B"
rocprofv3 --kernel-trace --stats -d $W/kt -o kt -- $B > $W/kt.log 2>&1
echo "# bench line under the profiler:"; grep "^{" $W/kt.log
python $R/tools/rocpd_summary.py $W/kt/kt_results.db | head -14
echo "# the library's own kernels:"
python $R/tools/rocpd_summary.py $W/kt/kt_results.db vibo
echo; echo "# one steady-state step of the replayed graph (tools/rocpd_sequence.py: dispatches between the last two launches of the ELBO kernel):"
echo "## fp32 rows (the headline step):"; python $R/tools/rocpd_sequence.py $W/kt/kt_results.db ELi0ELb0E; echo "## Format P rows:"; python $R/tools/rocpd_sequence.py $W/kt/kt_results.db ELi2ELb0E
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $pass | cut -d' ' -f1)
  echo; echo "# command: rocprofv3 --pmc $pass -- $B     (per-dispatch averages, vibo kernels only)"
  rocprofv3 --pmc $pass -d $W/$n -o p -- $B > /dev/null 2>&1
  python $R/tools/rocpd_summary.py $W/$n/p_results.db vibo | grep -E "ElboParams|msplit_kernel|split_kernel|finalize|epilogue|prologue"
done
echo; echo "# BASELINE configs[1] (100k x 1k, ability_dim 1) as the main workload of the same command:"
C2="python $R/bench.py --steps 20 --warmup 3 --persons 100000 --ability-dim 1 --no-cpu-baseline --no-extra --also-ability-dim 0 --no-also-config2 --no-format-p"
echo "# command: rocprofv3 --kernel-trace --stats -- $C2"
rocprofv3 --kernel-trace --stats -d $W/k2 -o kt -- $C2 > $W/k2.log 2>&1
grep "^{" $W/k2.log | cut -c1-400
python $R/tools/rocpd_summary.py $W/k2/kt_results.db vibo
python $R/tools/rocpd_sequence.py $W/k2/kt_results.db ELi0ELb0E
} > $S 2>&1
rm -rf $W
echo "wrote $S"
