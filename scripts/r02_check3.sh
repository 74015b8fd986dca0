#!/bin/bash
# baseline PMC of the 3D smoother (before the SELL kernel), cold-cache 2D smoother figures, new class tests
set -o pipefail
O=$(pwd)/gpurun_out/r02c
mkdir -p $O
REPO=$(pwd)
timeout -k 10 900 python -m pytest tests/test_solver_classes_gpu.py -m gpu -x -q > $O/class_tests.log 2>&1; echo "class tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/class_tests.log
timeout -k 10 300 python scripts/gpu_smoother_2d.py > $O/smoother_2d_cold.txt 2>&1; echo "smoother2d rc=$?" | tee -a $O/summary.txt
cat $O/smoother_2d_cold.txt
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc3d_$C -o tgv -- \
    python3 $REPO/bench.py --workload tgv3d-ipcs --cells 64 --steps 1 --warmup 1 > $O/pmc3d_$C.out 2> $O/pmc3d_$C.err; echo "pmc $C rc=$?" | tee -a $O/summary.txt
done
python3 $REPO/scripts/summarize_pmc.py $O/pmc3d_FETCH_SIZE $O/pmc3d_WRITE_SIZE > $O/r02_pre_tgv3d_n64_pmc_fetch_write_size.json
rm -rf $O/pmc3d_FETCH_SIZE $O/pmc3d_WRITE_SIZE
