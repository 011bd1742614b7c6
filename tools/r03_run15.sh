#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03m; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_sharded_gpu.py -x -q -k "native or hooks_sharded" > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
bash $R/tools/r03_cfg1_prof.sh > /dev/null 2>&1
ls $O $R/gpurun_out/r03cfg1
