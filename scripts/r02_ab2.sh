#!/bin/bash
O=$(pwd)/gpurun_out/r02m
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_3d.py tests/test_gpu_partition.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"
tail -3 $O/tests.log
for rep in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 50 > $O/bench_v2_$rep.json 2> $O/bench_v2_$rep.err
done
for v in 1 2; do
  NSFEM_STREAM_V=$v timeout -k 10 300 python bench.py --workload tgv3d-ipcs --cells 64 --steps 10 --warmup 3 > $O/tgv64_v$v.json 2> $O/tgv64_v$v.err
done
NSFEM_SELL=1 NSFEM_P2_ORDER=parity timeout -k 10 300 python bench.py --workload tgv3d-ipcs --cells 64 --steps 10 --warmup 3 > $O/tgv64_sell.json 2> $O/tgv64_sell.err
python scripts/show_bench.py $O/*.json
