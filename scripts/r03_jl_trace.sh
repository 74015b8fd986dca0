#!/bin/bash
# usage: scripts/r03_jl_trace.sh <tag> <n> [dbg values...]  -> gpurun_out/<tag>/jl_dbg<k>.txt (duration of the Jacobian kernels)
REPO=$(pwd); TAG=$1; N=$2; shift; shift; O=$REPO/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for D in "$@"; do
  echo "tile=${NSFEM_JL_TILE:-0}"
  export NSFEM_JL_DBG=$D
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$D -o b -- python3 $REPO/scripts/r03_jl_time.py $N > $O/run$D.log 2>&1 || exit 1
  python3 $REPO/scripts/summarize_trace_by_grid.py $(find $O/trace$D -name "*kernel_trace.csv" | head -1) 0 | grep -E "jac_lattice|conv_cell|spmv_dict" > $O/jl_dbg$D.txt
  rm -rf $O/trace$D
  echo "dbg=$D: $(cat $O/jl_dbg$D.txt)"
done
