#!/bin/bash
# One parameterised driver for the round's gpurun calls (run from the repository root on the GPU box):
#   tools/gpu_session.sh <out-dir-name> <stage> [<stage> ...]
# Stages write into gpurun_out/<out-dir-name>/; every stage is wrapped in its own timeout.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/$1; shift
mkdir -p "$O"
for stage in "$@"; do
  echo "=== stage $stage $(date +%T)"
  case $stage in
    seam)       timeout 900 python -m pytest tests/test_driver_seam.py tests/test_fullsize_gpu.py -m gpu -q --tb=short -p no:cacheprovider -s -k "driver or iid" 2>&1 | grep -v "^$" | tail -40 > $O/seam_tests.txt; grep -i "driver seam\|iid:\|passed\|failed\|Error\|assert" $O/seam_tests.txt | cut -c1-260 ;;
    il64ab)     for lib in "" il64_4_3 il64_8_4 il64_8_3; do echo "== lib=${lib:-default}" | tee -a $O/attn_il64_ab.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/attn_microbench.py 10,9216,5,64 25,4096,5,64 10,2304,10,64 25,1024,10,64 2>/dev/null | tee -a $O/attn_il64_ab.txt; done
                TOKENFLOW_HIP_LIB=build/variants/lib_il64_4_3.so timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_baseline_configs_gpu.py -q --tb=line -p no:cacheprovider -k "64 or cfg4 or cfg5" 2>&1 | tail -5 | tee -a $O/attn_il64_ab.txt ;;
    nosplitab)  for ns in 0 1; do echo "== TOKENFLOW_ATTN_NO_SPLIT=$ns" | tee -a $O/attn_nosplit_ab.txt
                  TOKENFLOW_ATTN_NO_SPLIT=$ns timeout 300 python tools/attn_microbench.py 8,256,8,160 8,64,8,160 4,256,8,80 4,64,8,160 10,144,20,64 2>/dev/null | tee -a $O/attn_nosplit_ab.txt; done ;;
    pmc40)      # instruction-level accounting of the level-0 kernel (no ATT decoder in this image: SQ counters, separate passes)
                for grp in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_IFETCH" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU_INT32" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_WAVES"; do
                  rm -rf /tmp/p40; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/p40 -- python $GRAFT_REPO_ROOT/tools/attn_microbench.py 8,4096,8,40 > /dev/null 2>>$GRAFT_REPO_ROOT/$O/pmc40.err )
                  DB=$(find /tmp/p40 -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/rocpd_pmc.py $DB | grep "il_kernel<BF16; 40; 8; 0\|^kernel" >> $O/pmc_l0_accounting.csv; done
                cut -c1-60,100-220 $O/pmc_l0_accounting.csv ;;
    r6ab)       # round 6: d = 64 / 40 / 80 streaming attention variants (tools/build_variants.sh), one process per library
                for lib in base "" il64a il64b il64c il64e base "" il64a; do echo "== lib=${lib:-default}" | tee -a $O/attn_d64_ab.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/attn_microbench.py 10,9216,5,64 25,4096,5,64 10,2304,10,64 25,1024,10,64 2>/dev/null | grep "inject=0" | tee -a $O/attn_d64_ab.txt; done
                for lib in "" il40p "" il40p; do echo "== lib=${lib:-default}" | tee -a $O/attn_d40_dma2_ab.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/attn_microbench.py 8,4096,8,40 4,1024,8,40 2>/dev/null | grep "inject=0" | tee -a $O/attn_d40_dma2_ab.txt; done
                for lib in "" il80a il80b "" il80a il80b; do echo "== lib=${lib:-default}" | tee -a $O/attn_d80_dma2_ab.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/attn_microbench.py 8,1024,8,80 4,256,8,80 2>/dev/null | grep "inject=0" | tee -a $O/attn_d80_dma2_ab.txt; done ;;
    r6abtests)  for lib in ${R6_TEST_LIBS:-"" il64a}; do echo "== lib=${lib:-default}" | tee -a $O/attn_variant_tests.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_baseline_configs_gpu.py tests/test_fullsize_gpu.py -q --tb=line -p no:cacheprovider -k "attn or cfg4 or cfg5 or cfg2" 2>&1 | tail -6 | tee -a $O/attn_variant_tests.txt; done ;;
    pmc64)      # instruction-level accounting of the d = 64 level-0 kernel at cfg4; PMC64_LIB selects the build
                for grp in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_IFETCH" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU_INT32" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_WAVES"; do
                  rm -rf /tmp/p64; ( cd /tmp && TOKENFLOW_HIP_LIB=${PMC64_LIB:+$GRAFT_REPO_ROOT/build/variants/lib_$PMC64_LIB.so} timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/p64 -- python $GRAFT_REPO_ROOT/tools/attn_microbench.py 10,9216,5,64 > /dev/null 2>>$GRAFT_REPO_ROOT/$O/pmc64.err )
                  DB=$(find /tmp/p64 -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/rocpd_pmc.py $DB | grep "pp_kernel<BF16; 64; 0\|il_kernel<BF16; 64; [48]; 0\|^kernel" >> $O/pmc_d64_${PMC64_LIB:-default}.csv; done
                cut -c1-60,100-220 $O/pmc_d64_${PMC64_LIB:-default}.csv ;;
    tailsplit)  # round 6: bank problems of large grids split into runs of frames where the last round of workgroups would run
                # nearly empty (split_plan's round model) against the unsplit launch
                for e in "TOKENFLOW_ATTN_NSEG=1" "X=0" "TOKENFLOW_ATTN_NSEG=2" "TOKENFLOW_ATTN_NSEG=5" "TOKENFLOW_ATTN_NSEG=1" "X=0"; do echo "== $e (X=0: the planner's choice)" | tee -a $O/attn_tail_split.txt
                  env $e timeout 300 python tools/attn_microbench.py 10,9216,5,64 10,2304,10,64 25,4096,5,64 8,1024,8,80 8,4096,8,40 2>/dev/null | grep "inject=0" | tee -a $O/attn_tail_split.txt; done ;;
    wiremodel)  for a in "" "--wire-model 25,50" "--wire-model 10,100"; do timeout 600 python tools/rank_step_microbench.py --native --only split,auto --no-levels --reps 12 $a 2>/dev/null | grep -v "^  wire:" | tee -a $O/rank_step_wire.txt; done
                timeout 300 python tools/rank_step_microbench.py --native --only split,auto --no-levels --no-copies --reps 12 2>/dev/null | tee -a $O/rank_step_wire.txt ;;
    hooksbd)    for a in "cfg2 6 --graph" "cfg2 10 --ranks 8 --wire-less --graph" "cfg2 6 --ranks 1 --graph" "cfg2 6 --ranks 1 --split --graph"; do timeout 900 python tools/hooks_bench.py $a --breakdown 2>&1 | grep -v amdgpu.ids | tee -a $O/hooks_breakdown.txt; done
                timeout 300 python tools/rank_step_microbench.py --native --only split,auto --no-copies --no-levels --reps 12 2>/dev/null | grep "step inject" | tee -a $O/hooks_breakdown.txt ;;
    r6ab2)      # round 6, second A/B: cross-phase prefetch, prefetch distance, static priority, denominator on the matrix pipe
                timeout 120 tools/ubench/mfma4_probe > $O/mfma4_probe.txt 2>&1; cat $O/mfma4_probe.txt
                for lib in noxpfvalu "" lsumvalu noxpf pf3 prio noxpfvalu ""; do echo "== lib=${lib:-default}" | tee -a $O/attn_il_ab2.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/attn_microbench.py 10,9216,5,64 25,4096,5,64 10,2304,10,64 8,4096,8,40 8,1024,8,80 2>/dev/null | grep "inject=0" | tee -a $O/attn_il_ab2.txt; done ;;
    d64tests)   timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_baseline_configs_gpu.py tests/test_fullsize_gpu.py -q --tb=short -p no:cacheprovider -k "attn or cfg4 or cfg5 or cfg2" 2>&1 | tail -8 | tee -a $O/d64_tests.txt ;;
    dualab)     for lib in "" dualdma dualdma4 dualdma8 "" dualdma; do echo "== lib=${lib:-default}" | tee -a $O/attn_d40_dual_ab.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/attn_microbench.py 8,4096,8,40 4,1024,8,40 2>/dev/null | grep "inject=1" | tee -a $O/attn_d40_dual_ab.txt; done ;;
    rb2ab)      for e in "TF_NN_RB2_MIN_WGS=100000000" "X=0" "TF_NN_RB2_MIN_WGS=100000000" "X=0"; do echo "== $e (X=0: two target tiles per wave where the plan takes them)" | tee -a $O/nn_rb2_ab.txt
                  env $e timeout 300 python tools/prop_microbench.py 8,5,4096,320 10,8,9216,320 25,8,4096,320 2>/dev/null | grep "one call" | tee -a $O/nn_rb2_ab.txt
                  env $e timeout 300 python tools/nn_microbench.py 8,5,4096,320 10,8,9216,320 2>/dev/null | tee -a $O/nn_rb2_ab.txt; done
                timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_kernels_gpu.py tests/test_baseline_configs_gpu.py -q --tb=short -p no:cacheprovider -k "nn_search or propagat or cfg2 or cfg4" 2>&1 | tail -4 | tee -a $O/nn_rb2_ab.txt ;;
    dual64ab)   for lib in "" il64dual8 il64dual4 "" il64dual8 il64dual4; do echo "== lib=${lib:-default}" | tee -a $O/attn_d64_dual_ab.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/attn_microbench.py 10,9216,5,64 10,2304,10,64 10,576,20,64 2>/dev/null | grep "inject=1" | tee -a $O/attn_d64_dual_ab.txt; done
                for lib in il64dual8 il64dual4; do TOKENFLOW_HIP_LIB=build/variants/lib_$lib.so timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_baseline_configs_gpu.py tests/test_fullsize_gpu.py -q --tb=short -p no:cacheprovider -k "attn or cfg4" 2>&1 | tail -4 | tee -a $O/attn_d64_dual_ab.txt; done ;;
    benchab)    # whole-step A/B on ONE box: round 5's kernel choices rebuilt from this tree (build/variants/lib_r05eq.so) against the default
                for lib in r05eq "" r05eq ""; do for cfg in cfg2 cfg4 cfg5; do
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 600 python bench.py --config $cfg --steps $([ $cfg = cfg2 ] && echo 10 || echo 2) --warmup 2 --no-cpu-baseline --no-yardstick --no-parity --no-other-configs > $O/benchab_tmp.json 2>> $O/benchab.err
                  python -c "import json;d=json.load(open('$O/benchab_tmp.json'));print('lib=${lib:-default}', '$cfg', 'ms/step', d['ms_per_step'], 'inject on/off', d['ms_per_step_inject_on'], d['ms_per_step_inject_off'], 'L0 attn ms', d['roofline']['avg_launch_ms'])" | tee -a $O/bench_step_ab.txt; done; done ;;
    bound80ab)  for lib in "" bound80 "" bound80; do echo "== lib=${lib:-default}" | tee -a $O/attn_d80_bound_ab.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/attn_microbench.py 8,1024,8,80 2>/dev/null | tee -a $O/attn_d80_bound_ab.txt; done ;;
    barrierexp) # timing experiments (WRONG results): the interleaved kernels without the per-tile barrier / without the DMA drain in front of it
                for lib in "" nobar nowait "" nobar nowait; do echo "== lib=${lib:-default}" | tee -a $O/attn_barrier_experiment.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/attn_microbench.py 10,9216,5,64 8,4096,8,40 8,1024,8,80 8,4096,8,80 2>/dev/null | grep "inject=0" | tee -a $O/attn_barrier_experiment.txt; done ;;
    rb2thr)     # threshold of the two-tile register-B search: the per-chunk searches of the hook path (one chunk per UNet pass) are small launches
                for e in "TF_NN_RB2_MIN_WGS=100000000" "TF_NN_RB2_MIN_WGS=2048" "TF_NN_RB2_MIN_WGS=1024" "TF_NN_RB2_MIN_WGS=512" "TF_NN_RB2_MIN_WGS=100000000" "TF_NN_RB2_MIN_WGS=1024"; do echo "== $e" | tee -a $O/nn_rb2_threshold.txt
                  env $e timeout 300 python tools/nn_microbench.py 8,5,4096,320 10,8,9216,320 25,8,4096,320 2>/dev/null | tee -a $O/nn_rb2_threshold.txt
                  env $e timeout 300 python tools/prop_microbench.py 8,5,4096,320 2>/dev/null | grep "per-chunk" | tee -a $O/nn_rb2_threshold.txt; done ;;
    dual80ab)   for lib in "" il80dual "" il80dual; do echo "== lib=${lib:-default}" | tee -a $O/attn_d80_dual_ab.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/attn_microbench.py 8,1024,8,80 10,576,20,64 2>/dev/null | grep "inject=1" | tee -a $O/attn_d80_dual_ab.txt; done
                TOKENFLOW_HIP_LIB=build/variants/lib_il80dual.so timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_baseline_configs_gpu.py tests/test_fullsize_gpu.py -q --tb=short -p no:cacheprovider -k "attn or cfg2" 2>&1 | tail -4 | tee -a $O/attn_d80_dual_ab.txt ;;
    seam2)      timeout 900 python -m pytest tests/test_driver_seam.py tests/test_sharded_gpu.py -m gpu -q --tb=short -p no:cacheprovider -s -k "driver or shard_vs_default" 2>&1 | grep -v "^$" | tail -60 > $O/seam2_tests.txt; grep -ai "driver seam\|passed\|failed\|Error\|assert" $O/seam2_tests.txt | cut -c1-300 ;;
    inputsab)   for n in 1 0 1 0; do timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-yardstick --no-parity --input-sets $n > $O/bench_sets_$n.json 2>> $O/inputsab.err; python -c "import json;d=json.load(open('$O/bench_sets_$n.json'));print('input sets',d['input_sets']['n'],d['ms_per_step'],d['ms_per_step_inject_on'],d['ms_per_step_inject_off'],d['roofline']['avg_launch_ms'])" | tee -a $O/input_sets_ab.txt; done ;;
    src4ab)     for lib in "" nosrc4 "" nosrc4; do echo "== lib=${lib:-current}" | tee -a $O/rank_step_src4_ab.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/rank_step_microbench.py --native --only split,auto --no-copies --reps 12 2>/dev/null | grep "step inject\|level 0" | tee -a $O/rank_step_src4_ab.txt; done ;;
    bmmprobe)   timeout 300 python tools/bmm_out_probe.py > $O/bmm_out_probe.txt 2>&1; grep -v amdgpu.ids $O/bmm_out_probe.txt | cut -c1-330 ;;
    srcaux)     for e in "X=0" "TOKENFLOW_RANK_SRC_AUX=1" "TOKENFLOW_RANK_SRC_AUX=1 TOKENFLOW_SPLIT_OVER=1" "TOKENFLOW_SPLIT_OVER=1" "X=0" "TOKENFLOW_RANK_SRC_AUX=1 TOKENFLOW_SPLIT_OVER=1"; do echo "== env $e" | tee -a $O/rank_step_srcaux_ab.txt
                  env $e timeout 300 python tools/rank_step_microbench.py --native --only split,auto --no-copies --reps 12 2>/dev/null | grep "step inject\|level 0" | tee -a $O/rank_step_srcaux_ab.txt; done ;;
    hooks2)     for a in "6" "6 --graph" "10 --ranks 8 --wire-less --graph" "10 --ranks 8 --wire-less"; do timeout 600 python tools/hooks_bench.py cfg2 $a >> $O/hooks_bench2.txt 2>/dev/null; done; cat $O/hooks_bench2.txt ;;
    hooksab)    for e in "TOKENFLOW_NORM1_ALL_BRANCHES=1" "TOKENFLOW_NORM1_ALL_BRANCHES=0" "TOKENFLOW_NORM1_ALL_BRANCHES=1" "TOKENFLOW_NORM1_ALL_BRANCHES=0"; do for a in "6 --graph" "10 --ranks 8 --wire-less --graph"; do echo -n "$e: " >> $O/hooks_norm1_ab.txt; env $e timeout 600 python tools/hooks_bench.py cfg2 $a >> $O/hooks_norm1_ab.txt 2>/dev/null; done; done; cat $O/hooks_norm1_ab.txt ;;
    onepassab)  for lib in "" il40nw4 "" il40nw4; do echo "== lib=${lib:-current}" | tee -a $O/rank_step_onepass_ab.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/rank_step_microbench.py --native --only onepass,auto --no-copies --reps 12 2>/dev/null | grep "step inject\|level 0" | tee -a $O/rank_step_onepass_ab.txt; done ;;
    gldstests)  timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_kernels_gpu.py tests/test_baseline_configs_gpu.py -q --tb=short -p no:cacheprovider -s -k "lds_dma or nn_search or propagat" 2>&1 | grep -v "^$" | tail -30 ;;
    gldsab)     for e in "TF_NN_GLDS_MIN_WGS=100000000" "TF_NN_GLDS_MIN_WGS=256" "TF_NN_GLDS_MIN_WGS=100000000" "TF_NN_GLDS_MIN_WGS=256" "TF_NN_GLDS_MIN_WGS=64"; do echo "== $e" | tee -a $O/nn_glds_ab.txt
                  env $e timeout 300 python tools/prop_microbench.py 8,5,1024,640 8,5,256,1280 10,8,2304,640 10,8,576,1280 25,8,1024,640 2>/dev/null | tee -a $O/nn_glds_ab.txt
                  env $e timeout 300 python tools/nn_microbench.py 8,5,1024,640 8,5,256,1280 2>/dev/null | tee -a $O/nn_glds_ab.txt; done ;;
    gldsrounds) for e in "TF_NN_GLDS_ROUNDS=3" "TF_NN_GLDS_ROUNDS=5" "TF_NN_GLDS_ROUNDS=9" "TF_NN_GLDS_ROUNDS=3" "TF_NN_GLDS_ROUNDS=5"; do echo "== $e" | tee -a $O/nn_glds_rounds.txt
                  env $e timeout 300 python tools/prop_microbench.py 8,5,1024,640 10,8,2304,640 10,8,576,1280 25,8,1024,640 25,8,256,1280 2>/dev/null | grep "one call" | tee -a $O/nn_glds_rounds.txt
                  env $e timeout 300 python tools/nn_microbench.py 8,5,1024,640 2>/dev/null | tee -a $O/nn_glds_rounds.txt; done ;;
    glds320)    for e in "TF_NN_GLDS_MIN_D=512" "TF_NN_GLDS_MIN_D=320" "TF_NN_GLDS_MIN_D=512" "TF_NN_GLDS_MIN_D=320"; do echo "== $e" | tee -a $O/nn_glds_d320.txt
                  env $e timeout 300 python tools/prop_microbench.py 8,5,4096,320 10,8,9216,320 4,2,1024,320 2>/dev/null | tee -a $O/nn_glds_d320.txt
                  env $e timeout 300 python tools/nn_microbench.py 8,5,4096,320 2>/dev/null | tee -a $O/nn_glds_d320.txt; done
                TF_NN_GLDS_MIN_D=320 timeout 600 python -m pytest tests/test_fullsize_gpu.py tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider -k "nn_search or propagat" 2>&1 | tail -3 | tee -a $O/nn_glds_d320.txt ;;
    hookstrace) rm -rf /tmp/ht; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ht -- python $GRAFT_REPO_ROOT/tools/hooks_bench.py cfg2 6 > /dev/null 2>&1 )
                python tools/rocpd_stats.py $(find /tmp/ht -name "*_results.db" | head -1) > $O/hooks_kernel_stats.csv; head -40 $O/hooks_kernel_stats.csv | cut -c1-200 ;;
    rbgab)      for e in "TF_NN_RB_GLDS=0" "TF_NN_RB_GLDS=1" "TF_NN_RB_GLDS=0" "TF_NN_RB_GLDS=1"; do echo "== $e" | tee -a $O/nn_rb_glds_ab.txt
                  env $e timeout 300 python tools/prop_microbench.py 8,5,4096,320 10,8,9216,320 2>/dev/null | grep "one call" | tee -a $O/nn_rb_glds_ab.txt
                  env $e timeout 300 python tools/nn_microbench.py 8,5,4096,320 10,8,9216,320 2>/dev/null | tee -a $O/nn_rb_glds_ab.txt; done
                TF_NN_RB_GLDS=1 timeout 600 python -m pytest tests/test_fullsize_gpu.py tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider -k "nn_search or propagat" 2>&1 | tail -3 | tee -a $O/nn_rb_glds_ab.txt ;;
    gldsspread) for lib in gldsblock "" gldsblock ""; do echo "== lib=${lib:-current (DMA pieces spread over the k-steps)}" | tee -a $O/nn_glds_spread_ab.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/prop_microbench.py 8,5,1024,640 10,8,2304,640 25,8,1024,640 10,8,576,1280 4,2,1024,320 2>/dev/null | grep "one call" | tee -a $O/nn_glds_spread_ab.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/nn_microbench.py 8,5,1024,640 2>/dev/null | tee -a $O/nn_glds_spread_ab.txt; done
                timeout 600 python -m pytest tests/test_fullsize_gpu.py tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider -k "lds_dma or nn_search or propagat" 2>&1 | tail -3 | tee -a $O/nn_glds_spread_ab.txt ;;
    il40dma)    for lib in "" il40dma "" il40dma; do echo "== lib=${lib:-current}" | tee -a $O/attn_il40_dma_ab.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/attn_microbench.py 8,4096,8,40 4,1024,8,40 2>/dev/null | tee -a $O/attn_il40_dma_ab.txt; done
                TOKENFLOW_HIP_LIB=build/variants/lib_il40dma.so timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_baseline_configs_gpu.py -q --tb=short -p no:cacheprovider -k "attn" 2>&1 | tail -6 | tee -a $O/attn_il40_dma_ab.txt ;;
    trprobe)    timeout 60 tools/ubench/tr_probe > $O/tr_probe.txt 2>&1; cat $O/tr_probe.txt ;;
    fusedtests) timeout 900 python -m pytest tests/test_fused_attn_gpu.py -q --tb=line -p no:cacheprovider 2>&1 | tail -40 > $O/fused_tests.txt; tail -25 $O/fused_tests.txt ;;
    kerneltests) timeout 1500 python -m pytest tests/test_kernels_gpu.py -q --tb=line -p no:cacheprovider -k "attn" 2>&1 | tail -30 > $O/kernel_attn_tests.txt; tail -15 $O/kernel_attn_tests.txt ;;
    shardtests) timeout 1500 python -m pytest tests/test_sharded_gpu.py -q --tb=short -p no:cacheprovider -x 2>&1 | tail -40 > $O/sharded_tests.txt; tail -15 $O/sharded_tests.txt ;;
    alltests)   timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -40 > $O/all_gpu_tests.txt; tail -15 $O/all_gpu_tests.txt ;;
    nntests)    timeout 900 python -m pytest tests/test_kernels_gpu.py -q --tb=line -p no:cacheprovider -k "nn_search or propagat" 2>&1 | tail -15 ;;
    abprev)     # A/B against build/variants/lib_prev.so (tools/build_variants.sh prev "<flags>")
                for lib in "" build/variants/lib_prev.so; do echo "--- lib: ${lib:-current}" | tee -a $O/ab_prev.txt
                  TOKENFLOW_HIP_LIB=$lib timeout 300 python tools/nn_microbench.py 8,5,256,1280 8,5,64,1280 4,4,64,1280 4,4,16,1280 2>/dev/null | tee -a $O/ab_prev.txt
                  TOKENFLOW_HIP_LIB=$lib timeout 300 python tools/fused_microbench.py --only "L1" --reps 10 2>/dev/null | cut -c1-330 | tee -a $O/ab_prev.txt; done ;;
    l1streamab) # round 6: a rank's level-1 pivotal pass in the streaming form (round-6 d = 80 kernels) against the fused launch
                for e in "X=0" "TOKENFLOW_FUSED_MAX_S=256" "X=0" "TOKENFLOW_FUSED_MAX_S=256"; do echo "== $e (X=0: fused launch at level 1)" | tee -a $O/rank_l1_stream_ab.txt
                  env $e timeout 600 python tools/rank_step_microbench.py --native --only split,auto --no-copies --reps 12 2>/dev/null | grep "step inject\|level 1" | tee -a $O/rank_l1_stream_ab.txt; done ;;
    il80w8ab)   # round 6: d = 80 plain kernel as 8-wave workgroups forced to 128 VGPRs (4 waves per SIMD, 19 spilled registers) against 4-wave / 145
                for lib in "" il80w8 "" il80w8; do echo "== lib=${lib:-default}" | tee -a $O/attn_d80_w8_ab.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/attn_microbench.py 8,1024,8,80 8,4096,8,80 4,256,8,80 2>/dev/null | grep "inject=0" | tee -a $O/attn_d80_w8_ab.txt; done ;;
    pmc80)      # instruction-level accounting of the d = 80 interleaved kernel: cfg2 level 1 and the same kernel on a level-0-sized grid
                for shape in 8,1024,8,80 8,4096,8,80; do
                for grp in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_IFETCH" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU_INT32" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_WAVES"; do
                  rm -rf /tmp/p80; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/p80 -- python $GRAFT_REPO_ROOT/tools/attn_microbench.py $shape > /dev/null 2>>$GRAFT_REPO_ROOT/$O/pmc80.err )
                  DB=$(find /tmp/p80 -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/rocpd_pmc.py $DB | grep "il_kernel<BF16; 80; 4; 0\|^kernel" >> $O/pmc_d80_S$(echo $shape | cut -d, -f2).csv; done
                python tools/l0_accounting.py $O/pmc_d80_S$(echo $shape | cut -d, -f2).csv | tee $O/d80_accounting_S$(echo $shape | cut -d, -f2).txt; done ;;
    mixab)      # round 6: mixed MFMA shapes in the d = 40 interleaved kernel (TF_TUNE_IL40_MIX: QK^T 32x32x16, P.V 16x16x32 behind v_permlane16_swap)
                TOKENFLOW_HIP_LIB=build/variants/lib_il40mix.so timeout 900 python -m pytest tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider -x -k "attn" 2>&1 | tail -12 | tee -a $O/attn_d40_mix_ab.txt
                for lib in "" il40mix "" il40mix "" il40mix; do echo "== lib=${lib:-default}" | tee -a $O/attn_d40_mix_ab.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/attn_microbench.py 8,4096,8,40 4,1024,8,40 2>/dev/null | tee -a $O/attn_d40_mix_ab.txt; done ;;
    mixab2)     for lib in "" il40mix il40mix2 il40mix3 "" il40mix il40mix2 il40mix3; do echo "== lib=${lib:-default}" | tee -a $O/attn_d40_mix_ab2.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/attn_microbench.py 8,4096,8,40 2>/dev/null | grep "inject=0" | tee -a $O/attn_d40_mix_ab2.txt; done ;;
    nosbab)     for lib in "" il40mix2 nosb40 nosb64 nosb80 "" il40mix2 nosb40 nosb64 nosb80; do echo "== lib=${lib:-default}" | tee -a $O/attn_il_nosb_ab.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/attn_microbench.py 8,4096,8,40 10,9216,5,64 8,1024,8,80 2>/dev/null | tee -a $O/attn_il_nosb_ab.txt; done ;;
    mixab3)     for lib in "" il40mix2 il40mix4 il40mix1o il40mix4o il40mix2o "" il40mix2 il40mix4 il40mix1o il40mix4o il40mix2o; do echo "== lib=${lib:-default}" | tee -a $O/attn_d40_mix_ab3.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/attn_microbench.py 8,4096,8,40 4,1024,8,40 2>/dev/null | grep "inject=0" | tee -a $O/attn_d40_mix_ab3.txt; done ;;
    mixstep)    # whole-step A/B on ONE box: the d = 40 kernel without the mixed MFMA shapes (build/variants/lib_nomix.so) against the default
                for lib in nomix "" nomix ""; do TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-yardstick --no-parity --no-other-configs > $O/bench_mix_${lib:-default}.json 2>> $O/mixstep.err
                  python -c "import json;d=json.load(open('$O/bench_mix_${lib:-default}.json'));print('lib=${lib:-default}', 'step', d['ms_per_step'], 'inject on/off', d['ms_per_step_inject_on'], d['ms_per_step_inject_off'], 'level-0 plain launch', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])" | tee -a $O/bench_mix_step_ab.txt; done ;;
    mixab4)     for lib in "" mixpf3 mixnoxpf mixpf1 "" mixpf3 mixnoxpf mixpf1; do echo "== lib=${lib:-default}" | tee -a $O/attn_d40_mix_ab4.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/attn_microbench.py 8,4096,8,40 2>/dev/null | grep "inject=0" | tee -a $O/attn_d40_mix_ab4.txt; done ;;
    mixab5)     for lib in "" mixw4m5 mixw4m4 "" mixw4m5 mixw4m4; do echo "== lib=${lib:-default}" | tee -a $O/attn_d40_mix_ab5.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/attn_microbench.py 8,4096,8,40 2>/dev/null | grep "inject=0" | tee -a $O/attn_d40_mix_ab5.txt; done ;;
    mixab6)     TOKENFLOW_HIP_LIB= timeout 900 python -m pytest tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider -x -k "attn" 2>&1 | tail -3 | tee -a $O/attn_d40_mix_ab6.txt
                for lib in mixnoswz "" mixnoswz "" mixnoswz ""; do echo "== lib=${lib:-default}" | tee -a $O/attn_d40_mix_ab6.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/attn_microbench.py 8,4096,8,40 4,1024,8,40 2>/dev/null | grep "inject=0" | tee -a $O/attn_d40_mix_ab6.txt; done
                rm -rf /tmp/sq; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES -d /tmp/sq -- python $GRAFT_REPO_ROOT/tools/attn_microbench.py 8,4096,8,40 > /dev/null 2>&1 )
                python tools/rocpd_pmc.py $(find /tmp/sq -name "*_results.db" | head -1) | grep "il_kernel<BF16; 40; 8; 0" | tee -a $O/attn_d40_mix_ab6.txt ;;
    mixrank)    for lib in nomix "" nomix ""; do echo "== lib=${lib:-default}" | tee -a $O/rank_step_mix_ab.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 600 python tools/rank_step_microbench.py --native --only split,auto --no-copies --reps 12 2>/dev/null | grep "step inject\|level 0" | tee -a $O/rank_step_mix_ab.txt; done ;;
    rbsab)      # round 6: the two-target-tile search with short MFMAs (nn_search_rbs_kernel, 16x16x32) against the 32x32x16 form (lib_norbs.so)
                timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_baseline_configs_gpu.py -q --tb=short -p no:cacheprovider -k "nn_search or propagat or iid or two_tile or cfg2 or cfg4" 2>&1 | tail -8 | tee -a $O/nn_rbs_ab.txt
                for lib in norbs "" norbs "" norbs ""; do echo "== lib=${lib:-default}" | tee -a $O/nn_rbs_ab.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/prop_microbench.py 8,5,4096,320 10,8,9216,320 25,8,4096,320 2>/dev/null | grep "one call" | tee -a $O/nn_rbs_ab.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/nn_microbench.py 8,5,4096,320 2>/dev/null | tee -a $O/nn_rbs_ab.txt; done ;;
    mix64ab)    # round 6: the mixed MFMA shapes at d = 64 (TF_TUNE_IL64_MIX): cfg4 / cfg5 level 0 and level 1; parity tests of the variant
                TOKENFLOW_HIP_LIB=build/variants/lib_mix64.so timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_baseline_configs_gpu.py -q --tb=short -p no:cacheprovider -k "64 or cfg4 or cfg5 or mixed" 2>&1 | tail -6 | tee -a $O/attn_d64_mix_ab.txt
                for lib in "" mix64 "" mix64; do echo "== lib=${lib:-default}" | tee -a $O/attn_d64_mix_ab.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/attn_microbench.py 10,9216,5,64 25,4096,5,64 10,2304,10,64 2>/dev/null | grep "inject=0" | tee -a $O/attn_d64_mix_ab.txt; done ;;
    gldsshab)   # round 6: the LDS-DMA search kernel (D >= 512; D = 320 with <= 1024 pivots) with short MFMAs against the 32x32x16 form (lib_nogldssh.so)
                timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_baseline_configs_gpu.py -q --tb=short -p no:cacheprovider -k "nn_search or propagat or iid or lds_dma or cfg1 or cfg2 or cfg4 or cfg5" 2>&1 | tail -6 | tee -a $O/nn_glds_sh_ab.txt
                for lib in nogldssh "" nogldssh "" nogldssh ""; do echo "== lib=${lib:-default}" | tee -a $O/nn_glds_sh_ab.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/prop_microbench.py 8,5,1024,640 10,8,2304,640 25,8,1024,640 8,5,256,1280 4,2,1024,320 2>/dev/null | grep "one call" | tee -a $O/nn_glds_sh_ab.txt
                  TOKENFLOW_HIP_LIB=${lib:+build/variants/lib_$lib.so} timeout 300 python tools/nn_microbench.py 8,5,1024,640 2>/dev/null | tee -a $O/nn_glds_sh_ab.txt; done ;;
    fusedbench) timeout 600 python tools/fused_microbench.py > $O/fused_microbench.txt 2>&1; tail -40 $O/fused_microbench.txt ;;
    rankstep)   timeout 600 python tools/rank_step_microbench.py --native --only split,auto > $O/rank_step_native.txt 2>&1; tail -14 $O/rank_step_native.txt
                timeout 300 python tools/rank_step_microbench.py --native --only split,auto --no-copies --no-levels > $O/rank_step_native_nocopies.txt 2>&1; tail -3 $O/rank_step_native_nocopies.txt
                timeout 300 python tools/rank_step_microbench.py --native --only onepass,auto --no-levels > $O/rank_step_native_onepass.txt 2>&1; tail -3 $O/rank_step_native_onepass.txt ;;
    bench)      timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python -c "import json;d=json.load(open('$O/bench.json'));print(d['ms_per_step'],d['value'],d['roofline']['frac'],d.get('parity',{}).get('attn_linf'),[ (l['attn_linf'],l['attn_linf_fp32_out']) for l in d.get('parity',{}).get('by_level',[])])" ;;
    benchquick) timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-yardstick > $O/bench_quick.json 2> $O/bench_quick.err; python -c "import json;d=json.load(open('$O/bench_quick.json'));print(d['ms_per_step'],d['value'],d['roofline']['frac'],[ (l['attn_linf'],l['attn_linf_fp32_out']) for l in d.get('parity',{}).get('by_level',[])])" ;;
    cfg1)       timeout 600 python bench.py --config cfg1 --steps 50 --warmup 10 --no-cpu-baseline --no-yardstick > $O/bench_cfg1.json 2> $O/bench_cfg1.err; python -c "import json;d=json.load(open('$O/bench_cfg1.json'));print('cfg1',d['ms_per_step'],[ (l['attn_linf'],l['attn_linf_fp32_out']) for l in d.get('parity',{}).get('by_level',[])])"
                timeout 600 python bench.py --config cfg1 --steps 50 --warmup 10 --graph --no-cpu-baseline --no-yardstick --no-parity > $O/bench_cfg1_graph.json 2> $O/bench_cfg1_graph.err; python -c "import json;d=json.load(open('$O/bench_cfg1_graph.json'));print('cfg1 graph',d['ms_per_step'])" ;;
    proxy)      timeout 300 tools/ubench/attn_loop_proxy > $O/attn_loop_proxy.txt 2>&1; cat $O/attn_loop_proxy.txt ;;
    ranktimeline) rm -rf /tmp/rt; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/rt -- python $GRAFT_REPO_ROOT/tools/rank_step_microbench.py --native --only split,auto --reps 6 --no-levels > /dev/null 2>$GRAFT_REPO_ROOT/$O/ranktimeline.err )
                DB=$(find /tmp/rt -name "*_results.db" | head -1); python tools/rocpd_timeline.py $DB --ms 5.6 > $O/rank_timeline.txt; python tools/rocpd_stats.py $DB > $O/rank_kernel_stats.csv; wc -l $O/rank_timeline.txt ;;
    ktrace)     rm -rf /tmp/kt; ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity --no-yardstick --steps 10 --warmup 2 > $GRAFT_REPO_ROOT/$O/bench_traced.json 2>/dev/null )
                python tools/rocpd_stats.py $(find /tmp/kt -name "*_results.db" | head -1) > $O/kernel_stats.csv; head -8 $O/kernel_stats.csv | cut -c1-200 ;;
    pmc)        for c in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/pmc_$c; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -- python $GRAFT_REPO_ROOT/tools/attn_microbench.py 8,4096,8,40 > /dev/null 2>&1 )
                  python tools/rocpd_pmc.py $(find /tmp/pmc_$c -name "*_results.db" | head -1) | grep -v "at::native" > $O/pmc_attn_$c.csv; done
                for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
                  rm -rf /tmp/sq; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $grp -d /tmp/sq -- python $GRAFT_REPO_ROOT/tools/attn_microbench.py 8,4096,8,40 8,256,8,160 > /dev/null 2>&1 )
                  python tools/rocpd_pmc.py $(find /tmp/sq -name "*_results.db" | head -1) | grep -v "at::native\|vt_pack" >> $O/pmc_attn_sq.csv; done
                head -3 $O/pmc_attn_FETCH_SIZE.csv | cut -c1-200 ;;
    cfg1trace)  rm -rf /tmp/kt1; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt1 -- python $GRAFT_REPO_ROOT/bench.py --config cfg1 --no-cpu-baseline --no-parity --no-yardstick --steps 40 --warmup 5 > /dev/null 2>&1 )
                python tools/rocpd_stats.py $(find /tmp/kt1 -name "*_results.db" | head -1) > $O/cfg1_kernel_stats.csv; head -30 $O/cfg1_kernel_stats.csv | cut -c1-190 ;;
    othercfgs)  for cfg in cfg4 cfg5; do timeout 600 python bench.py --config $cfg --no-cpu-baseline --no-yardstick --steps 3 --warmup 1 > $O/bench_$cfg.json 2>$O/bench_$cfg.err; python -c "import json;d=json.load(open('$O/bench_$cfg.json'));print('$cfg',d['ms_per_step'],d['parity']['attn_linf'],d['parity']['attn_linf_fp32_out'],d['parity']['nn_mismatch_rate'])"; done ;;
    hooks)      for a in "" "--graph" "--graph --all-chunks"; do timeout 600 python tools/hooks_bench.py cfg2 6 $a >> $O/hooks_bench.txt 2>/dev/null; done
                for a in "--ranks 8" "--ranks 8 --wire-less" "--ranks 8 --graph" "--ranks 8 --wire-less --graph"; do timeout 600 python tools/hooks_bench.py cfg2 10 $a >> $O/hooks_bench.txt 2>>$O/hooks_bench.err; done; cat $O/hooks_bench.txt ;;
    hookranks)  for a in "--ranks 8" "--ranks 8 --wire-less" "--ranks 8 --graph" "--ranks 8 --wire-less --graph"; do timeout 600 python tools/hooks_bench.py cfg2 10 $a >> $O/hooks_ranks.txt 2>>$O/hooks_ranks.err; done; cat $O/hooks_ranks.txt; grep -v amdgpu.ids $O/hooks_ranks.err | tail -20 ;;
    hookrankstrace) rm -rf /tmp/hrt; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/hrt -- python $GRAFT_REPO_ROOT/tools/hooks_bench.py cfg2 10 --ranks 8 --wire-less --graph > /dev/null 2>&1 )
                python tools/rocpd_stats.py $(find /tmp/hrt -name "*_results.db" | head -1) > $O/hooks_rank_kernel_stats.csv; head -40 $O/hooks_rank_kernel_stats.csv | cut -c1-170 ;;
    fusedpmc)   ( cd /tmp && rocprofv3 -L 2>/dev/null | grep -o "\(TCP\|TCC\|TA\|TD\|SQ\)_[A-Z0-9_a-z]*" | sort -u | tr "\n" " " | cut -c1-6000 ) > $O/pmc_counters_avail.txt
                for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TCC_REQ_sum"; do
                  rm -rf /tmp/fp; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/fp -- python $GRAFT_REPO_ROOT/tools/fused_pmc_target.py > /dev/null 2>>$GRAFT_REPO_ROOT/$O/fusedpmc.err )
                  DB=$(find /tmp/fp -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/rocpd_pmc.py $DB | grep -v "at::native" >> $O/pmc_fused.csv; done
                cat $O/pmc_fused.csv | cut -c1-200; tail -3 $O/fusedpmc.err ;;
    pmcl1)      # level-1 kernels: the d = 80 attention and the D = 640 8-chunk search
                for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
                  rm -rf /tmp/p1; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/p1 -- python $GRAFT_REPO_ROOT/tools/attn_microbench.py 8,1024,8,80 > /dev/null 2>&1 )
                  python tools/rocpd_pmc.py $(find /tmp/p1 -name "*_results.db" | head -1) | grep "ext_attn_il_kernel\|^kernel" >> $O/pmc_l1.csv
                  rm -rf /tmp/p2; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/p2 -- python $GRAFT_REPO_ROOT/tools/prop_microbench.py 8,5,1024,640 > /dev/null 2>&1 )
                  python tools/rocpd_pmc.py $(find /tmp/p2 -name "*_results.db" | head -1) | grep "nn_search_glds_kernel" >> $O/pmc_l1.csv; done
                cut -c1-60,100-220 $O/pmc_l1.csv ;;
    newtests)   timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "sharded_rank or loopback_transport or into_caller or nn_search_shapes" 2>&1 | tail -30 ;;
    hooktests)  timeout 1500 python -m pytest tests/test_baseline_configs_gpu.py tests/test_sharded_gpu.py -q --tb=short -p no:cacheprovider -x -k "hooks or hipgraph or cfg1" 2>&1 | tail -15 ;;
    gloo8)      timeout 900 python bench.py --gpus 8 --backend gloo --steps 2 --warmup 1 > $O/bench_gloo8.txt 2>&1; tail -c 1500 $O/bench_gloo8.txt ;;
    *) echo "unknown stage $stage" ;;
  esac
done
echo "=== done $(date +%T)"
