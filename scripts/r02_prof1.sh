#!/bin/bash
REPO=$(pwd); O=$REPO/gpurun_out/r02p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o b -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/trace.err
python3 $REPO/scripts/summarize_trace_by_grid.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 100 > $O/by_grid.csv
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/trace
head -5 $O/by_grid.csv
