#!/bin/bash
REPO=$(pwd); O=$REPO/gpurun_out/r02q3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o b -- python3 $REPO/bench.py --workload tgv3d-ipcs --cells 64 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/trace.err
python3 $REPO/scripts/summarize_trace_by_grid.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 0 > $O/by_grid.csv
rm -rf $O/trace
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace2 -o b -- python3 $REPO/bench.py --workload channel3d-bdf --cells 48 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_ch.json 2> $O/trace_ch.err
python3 $REPO/scripts/summarize_trace_by_grid.py $(find $O/trace2 -name "*kernel_trace.csv" | head -1) 0 > $O/by_grid_ch.csv
rm -rf $O/trace2
