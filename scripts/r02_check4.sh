#!/bin/bash
set -o pipefail
O=$(pwd)/gpurun_out/r02d
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/gpu_tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 env NSFEM_SELL=0 python bench.py --no-cpu-baseline > $O/bench_nosell.json 2> $O/bench_nosell.err; echo "bench nosell rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 python bench.py --workload tgv3d-ipcs --cells 64 --steps 10 --warmup 3 > $O/tgv64.json 2> $O/tgv64.err; echo "tgv rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 env NSFEM_SELL=0 python bench.py --workload tgv3d-ipcs --cells 64 --steps 10 --warmup 3 > $O/tgv64_nosell.json 2> $O/tgv64_nosell.err; echo "tgv nosell rc=$?" | tee -a $O/summary.txt
timeout -k 10 600 python bench.py --workload channel3d-bdf --cells 64 --steps 5 --warmup 3 > $O/ch3d_n64.json 2> $O/ch3d_n64.err; echo "ch3d n=64 rc=$?" | tee -a $O/summary.txt
