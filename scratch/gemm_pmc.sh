#!/bin/bash
# ON the GPU box: counter passes over single-shape GEMM launches, both tile variants
ROOT=$GRAFT_REPO_ROOT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
P3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_SALU"
for big in ${VARIANTS:-0 2}; do
for shp in "9472 1024 1024" "9472 3072 1024" "4096 4096 4096"; do
  t=$(echo $shp | tr ' ' 'x')
  export VMR_GEMM_BIG=$big
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    bash $ROOT/scratch/pmc_pass.sh gp_b${big}_${t}_p$i "$P" scratch/gemm_pmc.py $shp | grep -i "gemm" 
  done
done
done
