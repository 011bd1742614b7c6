#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03i; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_sharded_gpu.py -x -q > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
timeout 600 python bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "rc=$?" >> $O/bench_gloo2.err
timeout 900 python bench.py --gpus 8 --backend gloo --steps 2 --warmup 1 > $O/bench_gloo8.json 2> $O/bench_gloo8.err; echo "rc=$?" >> $O/bench_gloo8.err
ls $O
