#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03e; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_baseline_configs_gpu.py tests/test_sharded_gpu.py tests/test_kernels_gpu.py -x -q -k "fused or norm or propagate or hooks or sharded or comm or rccl or cfg1 or gather" > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
cd /tmp
timeout 300 python $R/tools/hooks_bench.py cfg2 6 > $O/hooks_bench.txt 2>/dev/null
TOKENFLOW_FUSED_GATHER_NORM=0 timeout 300 python $R/tools/hooks_bench.py cfg2 6 >> $O/hooks_bench.txt 2>/dev/null
timeout 300 python $R/tools/hooks_bench.py cfg2 6 --graph >> $O/hooks_bench.txt 2>/dev/null
timeout 300 python $R/tools/hooks_bench.py cfg2 6 --graph --all-chunks >> $O/hooks_bench.txt 2>/dev/null
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/tools/hooks_bench.py cfg2 6 > $O/hooks_traced.txt 2>/dev/null
python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*_results.db" | head -1) > $O/hooks_kernel_stats.csv
timeout 300 python $R/tools/rank_step_microbench.py --reps 10 --only split,heads --profile > $O/rank_step.txt 2>&1
timeout 300 python $R/tools/rank_step_microbench.py --reps 10 --only onepass,heads >> $O/rank_step.txt 2>&1
rm -rf /tmp/kt2; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt2 -- python $R/tools/rank_step_microbench.py --reps 5 --only split,heads > $O/rank_traced.txt 2>/dev/null
python $R/tools/rocpd_stats.py $(find /tmp/kt2 -name "*_results.db" | head -1) > $O/rank_kernel_stats.csv
timeout 300 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-yardstick > $O/bench.json 2> $O/bench.err
ls $O
