#!/bin/bash
cd /root/repo
O=gpurun_out/r02d; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_partition.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
run() { name=$1; shift; timeout -k 10 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; }; python scripts/show_bench.py $O/$name.json 2>/dev/null | head -30; }
run cavr2 --cells 256 --steps 10 --warmup 3 --local-ranks 2
NSFEM_DICT=0 run cavr2_csr --cells 256 --steps 10 --warmup 3 --local-ranks 2
