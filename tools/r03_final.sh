#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03final; mkdir -p $O
cd $R
timeout 3000 python -m pytest tests/ -q -m gpu > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
ls $O
