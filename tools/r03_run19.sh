#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03p; mkdir -p $O
timeout 300 python $R/tools/rccl_latency_probe.py > $O/rccl_probe.txt 2>&1
cd $R; timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -x -q -k "attn or split" > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
ls $O
