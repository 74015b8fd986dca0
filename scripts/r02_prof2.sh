#!/bin/bash
REPO=$(pwd); O=$REPO/gpurun_out/r02q
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o b -- python3 $REPO/bench.py --steps 40 --warmup 10 --timed-only > $O/bench.json 2> $O/trace.err
python3 $REPO/scripts/summarize_trace_by_grid.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 0 > $O/by_grid.csv
rm -rf $O/trace
cat $O/bench.json
