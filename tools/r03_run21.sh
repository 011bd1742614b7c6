#!/bin/bash
# final-build refresh of the kernel trace behind profiles/r03_kernel_stats.csv (same command as tools/r03_profile.sh)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py --no-cpu-baseline --no-parity --no-yardstick --steps 8 --warmup 2 > $O/bench_traced.json 2>/dev/null
python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*_results.db" | head -1) > $O/kernel_stats.csv
python $R/tools/attn_microbench.py > $O/attn_microbench.txt 2>&1
for a in "" "--graph" "--graph --all-chunks"; do python $R/tools/hooks_bench.py cfg2 6 $a >> $O/hooks_bench.txt 2>/dev/null; done
ls $O
