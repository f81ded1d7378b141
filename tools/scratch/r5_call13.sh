#!/bin/bash
# round 5: sharded cond/flow trainer tests, large-call replay test, REG-set flow backward on the primary panel only, bench line for profiles/
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_rccl.py "tests/test_gpu_trainer.py::test_fused_cond_flow_trainer_replays_bitwise_from_a_hipgraph" -m gpu -q -x > $O/r5_new13.log 2>&1; echo "pytest rc=$?" >> $O/r5_new13.log
tail -25 $O/r5_new13.log | cut -c1-300
S=$O/r5_ab13.txt; : > $S
for a in "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4" \
         "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4 --codes" \
         "--persons 1000000 --items 1000 --ability-dim 8 --flows 4" \
         "--persons 100000 --items 4096 --ability-dim 2 --flows 2"; do
  timeout 600 bash tools/ab_libs.sh "$a" c5 cur >> $S 2>&1
done
cat $S
timeout 1800 python -m pytest tests -m gpu -q > $O/r5_gpu_tests13.log 2>&1; echo "pytest rc=$?" >> $O/r5_gpu_tests13.log
grep -v '^\.' $O/r5_gpu_tests13.log | grep 'FAILED\|passed\|failed\|rc=' | head -40
timeout 900 python bench.py > $O/r5_bench13.log 2>&1; grep '^{' $O/r5_bench13.log > $O/r5_bench_line13.json; cut -c1-200 $O/r5_bench_line13.json
