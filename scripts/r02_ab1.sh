#!/bin/bash
O=$(pwd)/gpurun_out/r02l
mkdir -p $O
for rep in 1 2; do
  for v in 1 2; do
    NSFEM_STREAM_V=$v timeout -k 10 300 python bench.py --no-cpu-baseline --steps 50 > $O/bench_v${v}_$rep.json 2> $O/bench_v${v}_$rep.err
  done
done
for v in 1 2; do
  NSFEM_STREAM_V=$v timeout -k 10 300 python bench.py --workload tgv3d-ipcs --cells 64 --steps 10 --warmup 3 > $O/tgv64_v$v.json 2> $O/tgv64_v$v.err
done
NSFEM_SELL=1 NSFEM_P2_ORDER=parity timeout -k 10 300 python bench.py --workload tgv3d-ipcs --cells 64 --steps 10 --warmup 3 > $O/tgv64_sell.json 2> $O/tgv64_sell.err
python scripts/show_bench.py $O/*.json
