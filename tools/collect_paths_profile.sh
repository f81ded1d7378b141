#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace summaries of the non-headline kernel paths (tools/profile_kernel.py).
# Writes gpurun_out/paths_profile.txt (text only).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
S=$OUT/paths_profile.txt
: > $S
run() {
  rm -rf /tmp/kt
  echo "# rocprofv3 --kernel-trace --stats -- python tools/profile_kernel.py --iters 3 $*" >> $S
  rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/tools/profile_kernel.py --iters 3 $* > /tmp/kt.log 2>&1
  grep "terms/s" /tmp/kt.log >> $S
  python $R/tools/rocpd_summary.py /tmp/kt/kt_results.db vibo | cut -c1-150 >> $S
  echo >> $S
}
run --persons 1000000 --items 1000 --ability-dim 1 --irt 3
run --persons 1000000 --items 1000 --ability-dim 1 --flows 4
run --persons 1000000 --items 1000 --ability-dim 1 --cond
run --persons 100000 --items 10000 --ability-dim 1
run --persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4
# narrow rows (BASELINE configs[3] / [0] shapes): the narrow-row kernel (planner's choice), and the VALU row-split kernel it replaced
run --persons 535596 --items 96 --ability-dim 1 --missing 0.2
run --persons 535596 --items 96 --ability-dim 1 --missing 0.2 --kernel valu
run --persons 535596 --items 96 --ability-dim 4 --missing 0.2
run --persons 535596 --items 96 --ability-dim 1 --missing 0.2 --codes
run --persons 8000 --items 100 --ability-dim 1 --missing 0
run --persons 1000000 --items 1000 --ability-dim 8 --no-grad
# shuffled minibatch (rows through row_index), Format P cell codes, caller-supplied posterior (--ability-merge mean)
run --persons 1000000 --items 1000 --ability-dim 8 --gather
run --persons 1000000 --items 1000 --ability-dim 1 --gather
run --persons 1000000 --items 1000 --ability-dim 1 --codes
run --persons 1000000 --items 1000 --ability-dim 4 --codes
run --persons 1000000 --items 1000 --ability-dim 8 --codes --gather
run --persons 1000000 --items 1000 --ability-dim 1 --cond --codes
run --persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4 --codes
run --persons 1000000 --items 1000 --ability-dim 8 --given
rm -rf /tmp/kt
echo "wrote $S"
run --persons 1000000 --items 1000 --ability-dim 8 --cond
run --persons 1000000 --items 1000 --ability-dim 8 --flows 4
run --persons 1000000 --items 1000 --ability-dim 8 --irt 3
run --persons 200000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4
# the per-term MLP decoder kernel (tools/profile_decoder.py)
for m in deep link; do
  rm -rf /tmp/kt
  echo "# rocprofv3 --kernel-trace --stats -- python tools/profile_decoder.py --mode $m" >> $S
  rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/tools/profile_decoder.py --mode $m > /tmp/kt.log 2>&1
  grep "terms/s" /tmp/kt.log >> $S
  python $R/tools/rocpd_summary.py /tmp/kt/kt_results.db vibo | cut -c1-150 >> $S
  echo >> $S
done
