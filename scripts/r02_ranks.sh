#!/bin/bash
# rehearsal of the driver's N-rank bench commands with thread ranks on one GPU
cd /root/repo
O=gpurun_out/r02t; mkdir -p $O
run() { name=$1; shift; timeout -k 10 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -20 $O/$name.err; exit 1; }; python scripts/show_bench.py $O/$name.json 2>/dev/null | head -30; }
run w2_512 --gpus 2 --local-ranks 2 --steps 5 --warmup 2
run w4_512 --gpus 4 --local-ranks 4 --steps 3 --warmup 2
run w8_256 --gpus 8 --local-ranks 8 --cells 256 --steps 3 --warmup 2
run s8_960 --gpus 8 --local-ranks 8 --scaling strong --steps 3 --warmup 2
