#!/bin/bash
O=$(pwd)/gpurun_out/r02e
mkdir -p $O
for cfg in "2 512" "3 64"; do
  NSFEM_SELL=0 timeout -k 10 200 python scripts/gpu_sell_tune.py $cfg lex >> $O/tune.txt 2>&1
  NSFEM_SELL=0 timeout -k 10 200 python scripts/gpu_sell_tune.py $cfg parity >> $O/tune.txt 2>&1
  for v in 0 1 2 3; do
    NSFEM_SELL_VARIANT=$v timeout -k 10 200 python scripts/gpu_sell_tune.py $cfg parity >> $O/tune.txt 2>&1
  done
done
cat $O/tune.txt
