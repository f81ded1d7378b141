#!/bin/bash
# Run on the GPU box: the train step of ONE rank's shard of BASELINE configs[2] under 8-way person sharding (125 000 x 1 000,
# ability_dim 8) on a 1-rank nccl (RCCL) group -- the collective is captured into the step's second graph like on 8 ranks --
# as a rocprofv3 kernel trace: per-kernel averages and the dispatch sequence of one replayed step.
#   tools/collect_shard_profile.sh   -> gpurun_out/shard125k_step_sequence.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
W=/tmp/vibo_shard; rm -rf $W; mkdir -p $W
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --persons ${PERSONS:-125000} --force-dist --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-format-p --also-ability-dim 0 --no-also-config2 $BENCH_ARGS"
S=$OUT/shard125k_step_sequence.txt
{
echo "# bench line WITHOUT the profiler:  $B"
MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 $B 2>/dev/null | grep "^{"
echo "# command: rocprofv3 --kernel-trace --stats -- $B"
MASTER_ADDR=127.0.0.1 MASTER_PORT=29562 rocprofv3 --kernel-trace --stats -d $W/kt -o kt -- $B > $W/kt.log 2>&1
echo "# bench line under the profiler:"; grep "^{" $W/kt.log
python $R/tools/rocpd_summary.py $W/kt/kt_results.db | head -12
echo; echo "# one steady-state step of the replayed graphs (tools/rocpd_sequence.py: dispatches between two launches of the ELBO kernel):"
python $R/tools/rocpd_sequence.py $W/kt/kt_results.db msplit_kernel
} > $S 2>&1
rm -rf $W
cat $S
