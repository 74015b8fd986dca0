#!/bin/bash
O=$(pwd)/gpurun_out/r02u
mkdir -p $O
for rep in 1 2; do
for v in 0 1 2 3; do
  NSFEM_SELL_VARIANT=$v timeout -k 10 200 python scripts/gpu_sell_tune.py 3 64 parity >> $O/tune.txt 2>&1
done
done
cat $O/tune.txt
