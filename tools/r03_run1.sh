#!/bin/bash
# Round-3 GPU run 1: changed-kernel parity tests, rank-step microbench (W=8, wire-less), hooks-path kernel breakdown,
# 8-rank functional bench over gloo on one GPU, baseline bench.  Summaries under gpurun_out/r03/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_kernels_gpu.py "tests/test_baseline_configs_gpu.py" -x -q -k "sharded or split or strided or parts or interleav or ddim or rccl or c_abi or head" > $O/tests1.txt 2>&1; echo "tests rc=$?" >> $O/tests1.txt
cd /tmp
timeout 600 python $R/tools/rank_step_microbench.py --reps 10 > $O/rank_step.txt 2>&1
for lib in "" build/variants/lib_il384.so; do
  for ns in 0 1; do
    echo "== lib=${lib:-default} TOKENFLOW_ATTN_NO_SPLIT=$ns" >> $O/rank_shard.txt
    TOKENFLOW_HIP_LIB=${lib:+$R/$lib} TOKENFLOW_ATTN_NO_SPLIT=$ns timeout 300 python $R/tools/rank_shard_microbench.py >> $O/rank_shard.txt 2>&1
  done
done
for a in "" "--graph" "--graph --all-chunks"; do timeout 300 python $R/tools/hooks_bench.py cfg2 6 $a >> $O/hooks_bench.txt 2>/dev/null; done
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/tools/hooks_bench.py cfg2 6 > $O/hooks_traced.txt 2>/dev/null
python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*_results.db" | head -1) > $O/hooks_kernel_stats.csv
rm -rf /tmp/kt2; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt2 -- python $R/tools/hooks_bench.py cfg2 6 --graph > $O/hooks_traced_graph.txt 2>/dev/null
python $R/tools/rocpd_stats.py $(find /tmp/kt2 -name "*_results.db" | head -1) > $O/hooks_kernel_stats_graph.csv
timeout 600 python $R/bench.py --gpus 8 --backend gloo --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_gloo8.txt 2>&1
timeout 600 python $R/bench.py --gpus 8 --backend gloo --steps 2 --warmup 1 --no-cpu-baseline --config cfg5 >> $O/bench_gloo8.txt 2>&1
timeout 600 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
ls -la $O
