#!/bin/bash
# Collect the judged profiles of one round on the GPU box (run through gpurun from the repo root):
#   scripts/collect_profiles.sh r01_d
# 1. rocprofv3 --kernel-trace --stats of the default bench workload (kernel stats CSV + bench line)
# 2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of a 1-step run, summarised per kernel
# Outputs land in gpurun_out/<tag>_*; copy the summaries into profiles/ afterwards.
set -e
TAG=${1:-r01_x}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o ${TAG} -- \
  python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_trace.err
cp $(find $OUT/${TAG}_trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
python3 $REPO/scripts/summarize_trace_by_grid.py $(find $OUT/${TAG}_trace -name "*kernel_trace.csv" | head -1) 500 \
  > $OUT/${TAG}_kernel_stats_by_grid.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$C -o ${TAG} -- \
    python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/${TAG}_pmc_$C.err
done
python3 $REPO/scripts/summarize_pmc.py $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE > $OUT/${TAG}_pmc_fetch_write_size.json
tail -1 $OUT/${TAG}_bench.json
