#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03k; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -x -q -k "attn or inject or split or strided or bank" > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
cd /tmp
echo "== interleaved dual (default)" > $O/attn_ab.txt
timeout 300 python $R/tools/attn_microbench.py 8,4096,8,40 4,1024,8,40 10,9216,8,40 >> $O/attn_ab.txt 2>&1
echo "== plain dual (TF_TUNE_NO_IL40_DUAL)" >> $O/attn_ab.txt
TOKENFLOW_HIP_LIB=$R/build/variants/lib_noildual.so timeout 300 python $R/tools/attn_microbench.py 8,4096,8,40 4,1024,8,40 10,9216,8,40 >> $O/attn_ab.txt 2>&1
echo "== interleaved dual again" >> $O/attn_ab.txt
timeout 300 python $R/tools/attn_microbench.py 8,4096,8,40 >> $O/attn_ab.txt 2>&1
timeout 300 python $R/tools/rank_shard_microbench.py > $O/rank_shard.txt 2>&1
timeout 600 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-yardstick > $O/bench.json 2> $O/bench.err
ls $O
