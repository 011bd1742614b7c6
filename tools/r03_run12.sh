#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03l; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_sharded_gpu.py -x -q -k "hooks_sharded or native" > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
ls $O
