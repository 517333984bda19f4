#!/bin/bash
# instruction-mix / wait counters of the two stage kernels (separate --pmc passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf $R/gpurun_out/sq_r01; mkdir -p $R/gpurun_out/sq_r01
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_FMA_F64" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_BRANCH SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  RP_SKIP_SELF_CHECK=1 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/sq_r01/p$i -- python $R/scratch/pmc_run.py 64 4096 4 > /dev/null 2>&1
done
ls -R $R/gpurun_out/sq_r01 | grep counter_collection | head
