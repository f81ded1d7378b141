#!/bin/bash
# GPU box: per-kernel durations (rocprofv3 --kernel-trace) of tools/profile_kernel.py with the given flags.
#   tools/prof_paths.sh <tag> <profile_kernel flags...>    -> gpurun_out/paths_<tag>.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; shift
W=/tmp/vibo_paths_$tag; rm -rf $W; mkdir -p $W $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
{
echo "# command: rocprofv3 --kernel-trace --stats -- python tools/profile_kernel.py $*"
rocprofv3 --kernel-trace --stats -d $W -o kt -- python $R/tools/profile_kernel.py "$@" > $W/log 2>&1
tail -1 $W/log
python $R/tools/rocpd_summary.py $W/kt_results.db | head -12
} > $R/gpurun_out/paths_$tag.txt 2>&1
rm -rf $W
cat $R/gpurun_out/paths_$tag.txt
