#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03q; mkdir -p $O
cd $R; timeout 900 python -m pytest tests/test_sharded_gpu.py -x -q -k "native or hooks_sharded" > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
cd /tmp; timeout 300 python $R/tools/rank_step_microbench.py --reps 10 --only split,auto --native > $O/rank_step.txt 2>&1
ls $O
