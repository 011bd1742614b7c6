#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03o; mkdir -p $O
for lib in "" db160 sb160occ1; do
  echo "== lib=${lib:-default (single buffer, split to 2 waves/SIMD)}" >> $O/attn160.txt
  if [ -n "$lib" ]; then export TOKENFLOW_HIP_LIB=$R/build/variants/lib_$lib.so; else unset TOKENFLOW_HIP_LIB; fi
  timeout 300 python $R/tools/attn_microbench.py 8,256,8,160 8,64,8,160 4,64,8,160 20,576,20,64 >> $O/attn160.txt 2>&1
  timeout 300 python $R/tools/rank_shard_microbench.py 2>&1 | grep "level 2\|level 3" >> $O/attn160.txt
done
unset TOKENFLOW_HIP_LIB
cd $R
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -x -q -k "attn or split or strided or bank" > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
ls $O
