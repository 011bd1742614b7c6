#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03s; mkdir -p $O
timeout 600 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.err
timeout 600 python $R/bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-yardstick > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "rc=$?" >> $O/bench_cfg5.err
ls $O
