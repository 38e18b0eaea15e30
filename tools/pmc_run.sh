#!/bin/bash
# usage (on the GPU box): bash tools/pmc_run.sh <layer> <op> <tag>   -> gpurun_out/pmc_<tag>/p{1,2,3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc_$3
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_CVT SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $R/gpurun_out/pmc_$3/p$i -o r -- python3 $R/tools/pmc_layer.py $1 $2 > $R/gpurun_out/pmc_$3/log$i.txt 2>&1 || exit 1
  i=$((i+1))
done
