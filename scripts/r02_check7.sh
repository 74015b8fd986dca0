#!/bin/bash
set -o pipefail
O=$(pwd)/gpurun_out/r02s
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/gpu_tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 python bench.py --no-cpu-baseline --pressure-start previous > $O/bench_noextrap.json 2> $O/bench_noextrap.err
NSFEM_NO_FUSED_FIRST=1 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_nofusedfirst.json 2> $O/bench_nofusedfirst.err
timeout -k 10 300 python bench.py --workload tgv3d-ipcs --cells 64 --steps 10 --warmup 3 > $O/tgv64.json 2> $O/tgv64.err; echo "tgv rc=$?" | tee -a $O/summary.txt
python scripts/show_bench.py $O/*.json
