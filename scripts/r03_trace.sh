#!/bin/bash
# usage: scripts/r03_trace.sh <tag> [bench args...]   -> gpurun_out/<tag>/{bench.json,by_grid.csv}
REPO=$(pwd); TAG=$1; shift; O=$REPO/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o b -- python3 $REPO/bench.py --steps 40 --warmup 10 --timed-only "$@" > $O/bench.json 2> $O/trace.err
python3 $REPO/scripts/summarize_trace_by_grid.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 0 > $O/by_grid.csv
rm -rf $O/trace
tail -c 400 $O/bench.json
