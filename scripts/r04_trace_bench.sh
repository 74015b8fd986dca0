#!/bin/bash
# usage: scripts/r04_trace_bench.sh <tag> [bench args...]   -> gpurun_out/<tag>/{bench.json,by_grid.csv,gaps.txt}
REPO=$(pwd); TAG=$1; shift; O=$REPO/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o b -- python3 $REPO/bench.py --steps 40 --warmup 10 --timed-only "$@" > $O/bench.json 2> $O/trace.err
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 $REPO/scripts/summarize_trace_by_grid.py $T 0 > $O/by_grid.csv
python3 $REPO/scripts/trace_gaps.py $T 40 > $O/gaps.txt
rm -rf $O/trace
python3 $REPO/scripts/show_bench.py $O/bench.json; cat $O/gaps.txt
