#!/bin/bash
# SQ counters of k_jac_lattice: scripts/r03_pmc_jl.sh <tag> [n]   (NSFEM_JL_DBG exported by the caller)
REPO=$(pwd); TAG=$1; N=${2:-512}
export KERNEL=k_jac_lattice MINGRID=1000
scripts/pmc_kernel.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" ${TAG}_a scripts/r03_jl_time.py $N 5
scripts/pmc_kernel.sh "SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS" ${TAG}_b scripts/r03_jl_time.py $N 5
scripts/pmc_kernel.sh "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" ${TAG}_c scripts/r03_jl_time.py $N 5
