#!/bin/bash
O=$(pwd)/gpurun_out/r02g
mkdir -p $O
run() { timeout -k 10 200 python scripts/gpu_sell_tune.py "$@" >> $O/tune.txt 2>&1; }
for d in 0 1 2; do
  echo "NSFEM_SPMV_DEBUG=$d" >> $O/tune.txt
  NSFEM_SPMV_DEBUG=$d NSFEM_SELL=0 run 2 512 lex
  NSFEM_SPMV_DEBUG=$d NSFEM_SELL=0 run 2 1024 lex
  NSFEM_SPMV_DEBUG=$d NSFEM_SELL=0 run 3 64 lex
done
echo "NT=0" >> $O/tune.txt
NSFEM_SPMV_NT=0 NSFEM_SELL=0 run 2 1024 lex
NSFEM_SPMV_NT=0 NSFEM_SELL=0 run 3 64 lex
cat $O/tune.txt
