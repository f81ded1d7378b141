#!/bin/bash
# Run on the GPU box: the 16-person train step of the conditional-posterior / planar-flow models through the native
# trainer (FusedCondFlowTrainer: vibo_ctrain_prologue -> vibo_elbo_fwd_bwd -> vibo_ctrain_epilogue), eager and replayed from a
# hipGraph, next to the module + autograd + Adam step -- and the dispatch sequence of one step from a rocprofv3 kernel trace.
#   tools/collect_cond_step_profile.sh   -> gpurun_out/cond_step_sequence.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
S=$OUT/cond_step_sequence.txt
{
for cfg in "--irt 3 --cond --flows 4" "--irt 2 --cond --ability-dim 8" "--irt 2 --flows 2 --ability-dim 4"; do
  W=/tmp/vibo_cs; rm -rf $W; mkdir -p $W
  echo "# python tools/step_time.py --persons 20000 --items 1000 --batch 16 $cfg      (no profiler)"
  python $R/tools/step_time.py --persons 20000 --items 1000 --batch 16 $cfg 2>&1 | grep -v amdgpu.ids
  echo "# rocprofv3 --kernel-trace --stats -- (the same command): one replayed step of the native trainer"
  echo "# (under the tracer every dispatch is at least ~4.6 us; the row-index copy is the tool's own, not part of the step)"
  rocprofv3 --kernel-trace --stats -d $W/kt -o kt -- python $R/tools/step_time.py --persons 20000 --items 1000 --batch 16 $cfg > $W/kt.log 2>&1
  python $R/tools/rocpd_sequence.py $W/kt/kt_results.db ct_update_kernel | tail -25
  echo
done
} > $S 2>&1
rm -rf /tmp/vibo_cs
cat $S
