#!/bin/bash
# round-2 GPU check 1: new tests, bench line, channel3d-bdf convergence probe
set -o pipefail
O=gpurun_out/r02a
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/summary.txt
for n in 8 16 32; do
  timeout -k 10 300 python bench.py --workload channel3d-bdf --cells $n --steps 5 --warmup 3 > $O/ch3d_n$n.json 2> $O/ch3d_n$n.err; echo "ch3d n=$n rc=$?" | tee -a $O/summary.txt
done
tail -3 $O/gpu_tests.log
