#!/bin/bash
O=$(pwd)/gpurun_out/r02f
mkdir -p $O
run() { timeout -k 10 200 python scripts/gpu_sell_tune.py "$@" >> $O/tune.txt 2>&1; }
NSFEM_SELL=0 run 2 512 lex
for b in 8 16 64; do
  PARITY_BLOCK=$b NSFEM_SELL_VARIANT=1 run 2 512 parity
  PARITY_BLOCK=$b NSFEM_SELL_VARIANT=2 run 2 512 parity
done
PARITY_BLOCK=16 NSFEM_SELL_BALANCE=0 run 2 512 parity
PARITY_BLOCK=16 NSFEM_SELL=0 run 2 512 parity
NSFEM_SELL=0 run 3 64 lex
for b in 2 4 8; do
  PARITY_BLOCK=$b NSFEM_SELL_VARIANT=1 run 3 64 parity
  PARITY_BLOCK=$b NSFEM_SELL_VARIANT=2 run 3 64 parity
done
PARITY_BLOCK=4 NSFEM_SELL_VARIANT=0 run 3 64 parity
PARITY_BLOCK=4 NSFEM_SELL_VARIANT=3 run 3 64 parity
PARITY_BLOCK=4 NSFEM_SELL_BALANCE=0 run 3 64 parity
PARITY_BLOCK=4 NSFEM_SELL=0 run 3 64 parity
cat $O/tune.txt
