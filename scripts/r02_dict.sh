#!/bin/bash
cd /root/repo
O=gpurun_out/r02d; mkdir -p $O
run() { name=$1; shift; timeout -k 10 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; }; python scripts/show_bench.py $O/$name.json 2>/dev/null | head -30; }
run ch5 --workload channel3d-bdf --cells 48 --steps 5 --warmup 2 --no-cpu-baseline
run ch5_64 --workload channel3d-bdf --cells 64 --steps 5 --warmup 2 --no-cpu-baseline
run c3b5 --workload cavity3d-bdf --cells 32 --steps 5 --warmup 2 --no-cpu-baseline
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
