#!/bin/bash
# round 5: sticky 3PL saturation path, row counts a batch ahead, narrow-kernel unit size A/B, bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/r5_gpu_tests9.log 2>&1; echo "pytest rc=$?" >> $O/r5_gpu_tests9.log
grep -v '^\.' $O/r5_gpu_tests9.log | grep 'FAILED\|passed\|failed\|rc=' | head -40
S=$O/r5_ab9.txt; : > $S
for a in "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4" \
         "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4 --codes" \
         "--persons 1000000 --items 1000 --ability-dim 8 --irt 3" \
         "--persons 1000000 --items 1000 --ability-dim 8 --irt 3 --flows 4" \
         "--persons 1000000 --items 1000 --ability-dim 1 --irt 3 --item-scale 4" \
         "--persons 100000 --items 10000 --ability-dim 1" \
         "--persons 100000 --items 10000 --ability-dim 8"; do
  timeout 600 bash tools/ab_libs.sh "$a" c5 cur >> $S 2>&1
done
for a in "--persons 535596 --items 96 --ability-dim 1 --missing 0.2" "--persons 535596 --items 64 --ability-dim 1" "--persons 535596 --items 96 --ability-dim 4 --missing 0.2" "--persons 65536 --items 100 --ability-dim 1"; do
  timeout 600 bash tools/ab_libs.sh "$a" nwp1 cur nwp4 >> $S 2>&1
done
cat $S
timeout 900 python bench.py > $O/r5_bench9.log 2>&1; grep '^{' $O/r5_bench9.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'frac_step', d['roofline']['frac_step'])
print(json.dumps(d['extra']['other_shapes'])); print(json.dumps(d['extra']['config5_path'])[:300]); print(json.dumps(d['extra']['conditional_posterior']))"
