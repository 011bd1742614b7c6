#!/bin/bash
# Round-end evidence run on the GPU box: default bench, kernel trace of the same command, three PMC passes,
# micro-benchmarks, the other configs.  Writes summaries under gpurun_out/final/ (copied into profiles/ by hand).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
python $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace -d /tmp/kt -- python $R/bench.py --no-cpu-baseline --no-yardstick > $O/bench_traced.json 2>/dev/null
python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*_results.db" | head -1) > $O/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -- python $R/bench.py --no-cpu-baseline --no-yardstick --steps 2 --warmup 1 > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(find /tmp/pmc_$c -name "*_results.db" | head -1) > $O/pmc_$c.csv
done
python $R/tools/attn_microbench.py > $O/attn_microbench.txt 2>&1
python $R/tools/nn_microbench.py > $O/nn_microbench.txt 2>&1
for cfg in cfg1 cfg4 cfg5; do python $R/bench.py --config $cfg --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_$cfg.json 2>$O/bench_$cfg.err; done
ls -la $O
