#!/bin/bash
O=$(pwd)/gpurun_out/r02h
mkdir -p $O
run() { timeout -k 10 200 python scripts/gpu_sell_tune.py "$@" >> $O/tune.txt 2>&1; }
NSFEM_SELL=0 run 2 512 lex
for v in 0 1 2 3; do NSFEM_SELL_VARIANT=$v run 2 512 parity; done
NSFEM_SELL=0 run 2 1024 lex
for v in 1 2 3; do NSFEM_SELL_VARIANT=$v run 2 1024 parity; done
NSFEM_SELL=0 run 3 64 lex
for v in 0 1 2 3; do NSFEM_SELL_VARIANT=$v run 3 64 parity; done
cat $O/tune.txt
