#!/bin/bash
# Round-3 GPU run 2: parity of the new kernels (interleaved d = 80 / 160, register-B NN at D = 640), then A/B timings.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -x -q > $O/tests_kernels.txt 2>&1; echo "rc=$?" >> $O/tests_kernels.txt
timeout 600 python -m pytest tests/test_baseline_configs_gpu.py -x -q -k "ddim or cfg2 or levels or cfg1" > $O/tests_cfg.txt 2>&1; echo "rc=$?" >> $O/tests_cfg.txt
cd /tmp
for lib in "" build/variants/lib_base.so build/variants/lib_il80nw8.so build/variants/lib_occ160.so; do
  echo "== lib=${lib:-default}" >> $O/attn_ab.txt
  TOKENFLOW_HIP_LIB=${lib:+$R/$lib} timeout 300 python $R/tools/attn_microbench.py 8,1024,8,80 8,256,8,160 4,256,8,80 4,64,8,160 8,4096,8,40 >> $O/attn_ab.txt 2>&1
done
for lib in "" build/variants/lib_base.so; do
  echo "== lib=${lib:-default}" >> $O/nn_ab.txt
  TOKENFLOW_HIP_LIB=${lib:+$R/$lib} timeout 300 python $R/tools/nn_microbench.py 8,5,1024,640 8,5,4096,320 >> $O/nn_ab.txt 2>&1
  TOKENFLOW_HIP_LIB=${lib:+$R/$lib} timeout 300 python $R/tools/prop_microbench.py 8,5,1024,640 8,5,4096,320 4,2,256,640 >> $O/nn_ab.txt 2>&1
done
timeout 600 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
TOKENFLOW_HIP_LIB=$R/build/variants/lib_base.so timeout 600 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-yardstick > $O/bench_base.json 2>> $O/bench.err
timeout 300 python $R/bench.py --config cfg1 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_cfg1.json 2>> $O/bench.err
ls -la $O
