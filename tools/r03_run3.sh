#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_kernels_gpu.py -x -q -k "layer_norm or norm or injection_equals or split or attn" > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
cd /tmp
timeout 300 python $R/tools/ln_microbench.py > $O/ln_microbench.txt 2>&1
timeout 300 python $R/tools/hooks_bench.py cfg2 6 > $O/hooks_bench.txt 2>/dev/null
timeout 300 python $R/tools/hooks_bench.py cfg2 6 --graph --all-chunks >> $O/hooks_bench.txt 2>/dev/null
timeout 300 python $R/tools/attn_microbench.py 8,1024,8,80 8,4096,8,40 > $O/attn.txt 2>&1
timeout 300 python $R/tools/rank_step_microbench.py --reps 10 --only split,auto --profile > $O/rank_step_profile.txt 2>&1
timeout 300 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-yardstick > $O/bench.json 2> $O/bench.err
ls $O
