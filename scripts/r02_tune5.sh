#!/bin/bash
O=$(pwd)/gpurun_out/r02i
mkdir -p $O
run() { timeout -k 10 200 python scripts/gpu_sell_tune.py "$@" >> $O/tune.txt 2>&1; }
for rep in 1 2 3; do
  NSFEM_SELL=0 run 3 64 lex
  NSFEM_SELL_VARIANT=3 run 3 64 parity
  NSFEM_SELL_VARIANT=2 run 3 64 parity
done
for rep in 1 2 3; do
  NSFEM_SELL=0 run 2 512 lex
  NSFEM_SELL_VARIANT=1 run 2 512 parity
done
cat $O/tune.txt
