#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03cfg1; mkdir -p $O
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py --config cfg1 --no-cpu-baseline --no-parity --no-yardstick --steps 40 --warmup 5 > $O/bench_traced.json 2>/dev/null
python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*_results.db" | head -1) > $O/kernel_stats.csv
python $R/bench.py --config cfg1 --no-cpu-baseline --no-yardstick --steps 100 --warmup 10 > $O/bench.json 2>/dev/null
python $R/bench.py --config cfg1 --no-cpu-baseline --no-yardstick --steps 100 --warmup 10 --graph > $O/bench_graph.json 2>/dev/null
ls $O
