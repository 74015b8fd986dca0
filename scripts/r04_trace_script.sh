#!/bin/bash
# usage: scripts/r04_trace_script.sh <tag> <python script> [args...]   -> gpurun_out/<tag>/{out.log,by_grid.csv}
REPO=$(pwd); TAG=$1; shift; O=$REPO/gpurun_out/$TAG
mkdir -p $O
SCRIPT=$REPO/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o b -- python3 $SCRIPT "$@" > $O/out.log 2> $O/trace.err
python3 $REPO/scripts/summarize_trace_by_grid.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 0 > $O/by_grid.csv
rm -rf $O/trace
tail -c 600 $O/out.log
