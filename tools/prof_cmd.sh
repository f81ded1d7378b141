#!/bin/bash
# GPU box: per-kernel durations (rocprofv3 --kernel-trace) of an arbitrary command.
#   tools/prof_cmd.sh <tag> <command...>    -> gpurun_out/cmd_<tag>.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; shift
W=/tmp/vibo_cmd_$tag; rm -rf $W; mkdir -p $W $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
{
echo "# command: rocprofv3 --kernel-trace --stats -- $*"
(cd $R && rocprofv3 --kernel-trace --stats -d $W -o kt -- "$@") > $W/log 2>&1
grep -v "^W2026\|amdgpu.ids" $W/log | tail -4
python $R/tools/rocpd_summary.py $W/kt_results.db | head -${TOP:-45}
} > $R/gpurun_out/cmd_$tag.txt 2>&1
rm -rf $W
cat $R/gpurun_out/cmd_$tag.txt
