#!/bin/bash
O=$(pwd)/gpurun_out/r02v
mkdir -p $O
for rep in 1 2; do
  NSFEM_SELL=0 timeout -k 10 200 python scripts/gpu_sell_tune.py 2 512 lex >> $O/tune.txt 2>&1
  NSFEM_SELL=2 timeout -k 10 200 python scripts/gpu_sell_tune.py 2 512 parity >> $O/tune.txt 2>&1
done
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 50 > $O/bench_lex.json 2> $O/bench_lex.err
NSFEM_SELL=2 NSFEM_P2_ORDER=parity timeout -k 10 300 python bench.py --no-cpu-baseline --steps 50 > $O/bench_sell.json 2> $O/bench_sell.err
cat $O/tune.txt; python scripts/show_bench.py $O/*.json
