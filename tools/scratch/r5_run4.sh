#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python tools/scratch/r5_dbg.py 2>&1 | grep '^(\|bad entries' > $O/r5_dbg4.txt; cat $O/r5_dbg4.txt
timeout 1200 python -m pytest tests -m gpu -q > $O/r5_gpu_tests4.log 2>&1; echo "pytest rc=$?" >> $O/r5_gpu_tests4.log
grep -v '^\.' $O/r5_gpu_tests4.log | grep 'FAILED\|passed\|failed\|rc=' | head -40
S=$O/r5_ab4.txt; : > $S
for a in "--persons 100000 --items 10000 --ability-dim 1 --irt 3 --cond --flows 4" \
         "--persons 100000 --items 10000 --ability-dim 1" \
         "--persons 1000000 --items 1000 --ability-dim 1 --irt 3" \
         "--persons 1000000 --items 1000 --ability-dim 8" \
         "--persons 1000000 --items 1000 --ability-dim 1" \
         "--persons 125000 --items 1000 --ability-dim 8"; do
  timeout 600 bash tools/ab_libs.sh "$a" r4 cur >> $S 2>&1
done
cat $S
python tools/profile_decoder.py --mode deep 2>&1 | tail -2
VIBO_HIP_LIB=$R/variational-item-response-theory-public_amd/vibo_amd/variants/libvibo_r4.so python tools/profile_decoder.py --mode deep 2>&1 | tail -2
