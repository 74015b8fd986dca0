#!/bin/bash
# rehearsal of the N-rank bench paths on one GPU (thread ranks, in-process communicator)
cd /root/repo
O=gpurun_out/r02t; mkdir -p $O
run() { name=$1; shift; timeout -k 10 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "FAILED $name"; tail -20 $O/$name.err; exit 1; }; python scripts/show_bench.py $O/$name.json 2>/dev/null | head -30; }
run ch_n16_1 --workload channel3d-bdf --cells 16 --steps 5 --warmup 2 --no-cpu-baseline
run ch_n16_2strong --workload channel3d-bdf --cells 16 --steps 5 --warmup 2 --local-ranks 2 --scaling strong
run ch_n16_4strong --workload channel3d-bdf --cells 16 --steps 5 --warmup 2 --local-ranks 4 --scaling strong
run ch_n16_2weak --workload channel3d-bdf --cells 16 --steps 5 --warmup 2 --local-ranks 2
run cav_2 --cells 128 --steps 5 --warmup 2 --local-ranks 2
run cav_4s --cells 256 --steps 5 --warmup 2 --local-ranks 4 --scaling strong
run tgv_2 --workload tgv3d-ipcs --cells 16 --steps 5 --warmup 2 --local-ranks 2
run c3d_2 --workload cavity3d-ipcs --cells 16 --steps 5 --warmup 2 --local-ranks 2
