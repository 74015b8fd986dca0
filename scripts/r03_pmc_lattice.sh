#!/bin/bash
# SQ counters of the lattice kernel's finest-level launches: scripts/r03_pmc_lattice.sh <tag>  (env switches exported by the caller)
REPO=$(pwd); TAG=$1
export NSFEM_SWEEP_CHILD=1 KERNEL=${KERNEL:-k_cheb_lattice} MINGRID=${MINGRID:-100000}
scripts/pmc_kernel.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" ${TAG}_a scripts/gpu_lattice_sweep.py 512
scripts/pmc_kernel.sh "SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS" ${TAG}_b scripts/gpu_lattice_sweep.py 512
scripts/pmc_kernel.sh "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC" ${TAG}_c scripts/gpu_lattice_sweep.py 512
