#!/bin/bash
O=$(pwd)/gpurun_out/r02j
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_3d.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tune.txt
tail -3 $O/tests.log >> $O/tune.txt
run() { timeout -k 10 200 python scripts/gpu_sell_tune.py "$@" >> $O/tune.txt 2>&1; }
for rep in 1 2; do
  NSFEM_SELL=0 run 3 64 lex
  NSFEM_SELL_VARIANT=3 run 3 64 parity
  NSFEM_SELL=0 run 2 512 lex
  NSFEM_SELL=0 run 2 1024 lex
done
NSFEM_SPMV_DEBUG=2 NSFEM_SELL=0 run 3 64 lex
NSFEM_SPMV_DEBUG=2 NSFEM_SELL=0 run 2 1024 lex
cat $O/tune.txt
