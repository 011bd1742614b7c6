#!/bin/bash
# Round-3 evidence run on the GPU box: default bench, kernel trace of the same command, PMC passes (separate runs,
# --kernel-trace + --pmc only), micro-benchmarks, the other configs, hook path, one rank of 8.  Summaries under
# gpurun_out/r03/ (copied into profiles/r03_* by hand).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py --no-cpu-baseline --no-parity --no-yardstick --steps 8 --warmup 2 > $O/bench_traced.json 2>/dev/null
python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*_results.db" | head -1) > $O/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  rm -rf /tmp/pmc_$c; rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -- python $R/tools/attn_microbench.py 8,4096,8,40 8,1024,8,80 > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(find /tmp/pmc_$c -name "*_results.db" | head -1) | grep -v "at::native" > $O/pmc_attn_$c.csv
done
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  rm -rf /tmp/sq; rocprofv3 --kernel-trace --pmc $grp -d /tmp/sq -- python $R/tools/attn_microbench.py 8,4096,8,40 8,1024,8,80 > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(find /tmp/sq -name "*_results.db" | head -1) | grep -v "at::native\|vt_pack" >> $O/pmc_attn_sq.csv
done
python $R/tools/attn_microbench.py > $O/attn_microbench.txt 2>&1
python $R/tools/prop_microbench.py > $O/prop_microbench.txt 2>&1
for cfg in cfg1 cfg4 cfg5; do python $R/bench.py --config $cfg --no-cpu-baseline --no-yardstick --steps 3 --warmup 1 > $O/bench_$cfg.json 2>$O/bench_$cfg.err; done
python $R/bench.py --config cfg1 --no-cpu-baseline --no-yardstick --steps 50 --warmup 5 --graph > $O/bench_cfg1_graph.json 2>>$O/bench_cfg1.err
for a in "" "--graph" "--graph --all-chunks" "--proj"; do python $R/tools/hooks_bench.py cfg2 6 $a >> $O/hooks_bench.txt 2>/dev/null; done
python $R/tools/rank_step_microbench.py --reps 10 --only split,auto --native > $O/rank_step.txt 2>&1
python $R/tools/rank_step_microbench.py --reps 10 --only split,auto >> $O/rank_step.txt 2>&1
rm -rf /tmp/kt2; rocprofv3 --kernel-trace --stats -d /tmp/kt2 -- python $R/tools/rank_step_microbench.py --reps 5 --only split,auto --native > $O/rank_traced.txt 2>/dev/null
python $R/tools/rocpd_stats.py $(find /tmp/kt2 -name "*_results.db" | head -1) > $O/rank_kernel_stats.csv
ls -la $O
