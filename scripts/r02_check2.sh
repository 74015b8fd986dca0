#!/bin/bash
set -o pipefail
O=gpurun_out/r02b
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/gpu_tests.log
timeout -k 10 600 python bench.py --workload channel3d-bdf --cells 64 --steps 5 --warmup 3 > $O/ch3d_n64.json 2> $O/ch3d_n64.err; echo "ch3d n=64 rc=$?" | tee -a $O/summary.txt
