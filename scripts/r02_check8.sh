#!/bin/bash
set -o pipefail
O=$(pwd)/gpurun_out/r02t
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_3d.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/tests.log
timeout -k 10 300 python bench.py --workload tgv3d-ipcs --cells 64 --steps 10 --warmup 3 > $O/tgv64.json 2> $O/tgv64.err; echo "tgv rc=$?" | tee -a $O/summary.txt
timeout -k 10 600 python bench.py --workload channel3d-bdf --cells 48 --steps 3 --warmup 2 > $O/ch48.json 2> $O/ch48.err; echo "ch rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 python scripts/gpu_sell_tune.py 3 64 parity > $O/sell_cold.txt 2>&1
python scripts/show_bench.py $O/*.json; cat $O/sell_cold.txt
