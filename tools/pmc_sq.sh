#!/bin/bash
# SQ counter passes (separate runs, --kernel-trace only) over the two micro-benchmarks; summaries to gpurun_out/sq/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sq; mkdir -p $O
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_INSTS_MFMA"; do
  i=$((i+1))
  for b in attn nn; do
    if [ $b = attn ]; then cmd="python $R/tools/attn_microbench.py 8,4096,8,40 8,1024,8,80"; else cmd="python $R/tools/nn_microbench.py 8,5,4096,320 8,5,1024,640"; fi
    rm -rf /tmp/sq; rocprofv3 --kernel-trace --pmc $grp -d /tmp/sq -- $cmd > /dev/null 2>&1
    DB=$(find /tmp/sq -name "*_results.db" | head -1)
    [ -n "$DB" ] && python $R/tools/rocpd_pmc.py $DB | grep -v "at::native\|vt_pack\|inv_norm\|finalize" >> $O/${b}_sq.csv
  done
done
wc -l $O/*.csv
