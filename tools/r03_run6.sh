#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_baseline_configs_gpu.py tests/test_sharded_gpu.py tests/test_kernels_gpu.py -x -q -k "fused_norm or sharded or layer_norm" > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
cd /tmp
timeout 300 python $R/tools/rank_step_microbench.py --reps 10 --only split,heads --profile > $O/rank_step.txt 2>&1
timeout 300 python $R/tools/rank_step_microbench.py --reps 10 --only split,auto >> $O/rank_step.txt 2>&1
timeout 300 python $R/tools/rank_step_microbench.py --reps 10 --only onepass,heads >> $O/rank_step.txt 2>&1
timeout 300 python $R/tools/rank_step_microbench.py --reps 10 --only split,bank >> $O/rank_step.txt 2>&1
ls $O
