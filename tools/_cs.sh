cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "--irt 3 --cond --flows 4" "--irt 2 --cond --ability-dim 8" "--irt 2 --flows 2 --ability-dim 4"; do
W=/tmp/cs; rm -rf $W; mkdir -p $W
python $R/tools/step_time.py --persons 20000 --items 1000 --batch 16 $cfg 2>&1 | grep -v amdgpu.ids
rocprofv3 --kernel-trace --stats -d $W/kt -o kt -- python $R/tools/step_time.py --persons 20000 --items 1000 --batch 16 $cfg > $W/kt.log 2>&1
python $R/tools/rocpd_sequence.py $W/kt/kt_results.db ct_finish_kernel | tail -25
done
