#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03d; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_baseline_configs_gpu.py tests/test_sharded_gpu.py -x -q -k "fused or norm or propagate or hooks or sharded or comm or rccl or cfg1" > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
cd /tmp
timeout 300 python $R/tools/hooks_bench.py cfg2 6 > $O/hooks_bench.txt 2>/dev/null
TOKENFLOW_FUSED_GATHER_NORM=0 timeout 300 python $R/tools/hooks_bench.py cfg2 6 >> $O/hooks_bench.txt 2>/dev/null
timeout 300 python $R/tools/hooks_bench.py cfg2 6 --graph >> $O/hooks_bench.txt 2>/dev/null
timeout 300 python $R/tools/hooks_bench.py cfg2 6 --graph --all-chunks >> $O/hooks_bench.txt 2>/dev/null
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/tools/hooks_bench.py cfg2 6 > $O/hooks_traced.txt 2>/dev/null
python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*_results.db" | head -1) > $O/hooks_kernel_stats.csv
timeout 300 python $R/tools/rank_step_microbench.py --reps 10 --only split,auto > $O/rank_step.txt 2>&1
timeout 300 python $R/tools/rank_step_microbench.py --reps 10 --only onepass,auto >> $O/rank_step.txt 2>&1
timeout 300 python $R/tools/rank_step_microbench.py --reps 10 --only split,heads --profile >> $O/rank_step.txt 2>&1
ls $O
