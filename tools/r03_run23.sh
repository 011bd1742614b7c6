#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03t; mkdir -p $O
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/sq; rocprofv3 --kernel-trace --pmc $grp -d /tmp/sq -- python $R/tools/attn_microbench.py 8,4096,8,40 8,256,8,160 > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(find /tmp/sq -name "*_results.db" | head -1) | grep -v "at::native\|vt_pack" >> $O/pmc_attn_sq_final.csv
done
ls $O
