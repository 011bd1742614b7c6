#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03n; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_sharded_gpu.py -x -q > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "rc=$?" >> $O/smoke.txt
cd /tmp
timeout 300 python $R/tools/rank_step_microbench.py --reps 10 --only split,auto --native > $O/rank_step.txt 2>&1
ls $O
