#!/bin/bash
cd /root/repo
O=gpurun_out/r02d; mkdir -p $O
run() { name=$1; shift; timeout -k 10 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -5 $O/$name.err; }; python scripts/show_bench.py $O/$name.json 2>/dev/null | head -30; }
run dfg3 --workload dfg-bdf --steps 10 --warmup 3 --no-cpu-baseline
run c3d --workload cavity3d-ipcs --cells 32 --steps 5 --warmup 2 --no-cpu-baseline
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
