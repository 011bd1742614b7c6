#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03g; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_sharded_gpu.py -x -q > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
timeout 600 python -m pytest tests/test_baseline_configs_gpu.py -x -q -k "cfg1 or split" >> $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
cd /tmp
echo "== ticket merge (default)" > $O/rank_shard.txt
timeout 300 python $R/tools/rank_shard_microbench.py >> $O/rank_shard.txt 2>&1
echo "== merge launch (variant)" >> $O/rank_shard.txt
TOKENFLOW_HIP_LIB=$R/build/variants/lib_mergelaunch.so timeout 300 python $R/tools/rank_shard_microbench.py >> $O/rank_shard.txt 2>&1
timeout 300 python $R/tools/rank_step_microbench.py --reps 10 --only split,heads > $O/rank_step.txt 2>&1
TOKENFLOW_SHARD_SRC_AUX=0 timeout 300 python $R/tools/rank_step_microbench.py --reps 10 --only split,heads >> $O/rank_step.txt 2>&1
timeout 300 python $R/tools/rank_step_microbench.py --reps 10 --only onepass,heads >> $O/rank_step.txt 2>&1
timeout 300 python $R/bench.py --config cfg1 --steps 50 --warmup 10 --no-cpu-baseline --no-yardstick > $O/bench_cfg1.json 2> $O/bench_cfg1.err
timeout 300 python $R/bench.py --config cfg1 --graph --steps 50 --warmup 10 --no-cpu-baseline --no-yardstick > $O/bench_cfg1_graph.json 2>> $O/bench_cfg1.err
TOKENFLOW_HIP_LIB=$R/build/variants/lib_mergelaunch.so timeout 300 python $R/bench.py --config cfg1 --steps 50 --warmup 10 --no-cpu-baseline --no-yardstick --no-parity > $O/bench_cfg1_mergelaunch.json 2>> $O/bench_cfg1.err
ls $O
