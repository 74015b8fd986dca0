#!/bin/bash
O=$(pwd)/gpurun_out/r02r
mkdir -p $O
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 60 --warmup 5 --timed-only > $O/$name.json 2> $O/$name.err; }
run base A=1
run post2 NSFEM_MGV_POST=2
run post4 NSFEM_MGV_POST=4
run pre1post2 NSFEM_MGV_PRE=1 NSFEM_MGV_POST=2
for t in "2,0.1" "8,0.1" "4,0.3" "16,0.2"; do
  timeout -k 10 200 python bench.py --steps 60 --warmup 5 --timed-only --mg-truncation $t > $O/trunc_$t.json 2> $O/trunc_$t.err
done
timeout -k 10 200 python bench.py --steps 60 --warmup 5 --timed-only --mg-eig-ratio 3 > $O/ratio3.json 2>/dev/null
timeout -k 10 200 python bench.py --steps 60 --warmup 5 --timed-only --mg-eig-ratio 6 > $O/ratio6.json 2>/dev/null
timeout -k 10 200 python bench.py --steps 60 --warmup 5 --timed-only --mass-solver cg > $O/masscg.json 2>/dev/null
python scripts/show_bench.py $O/*.json
