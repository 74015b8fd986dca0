#!/bin/bash
# per-phase cost of the leg kernels: traces with the first k operations only (knock-out build)
for k in 0 1 3 5 6 7 8 9 10 12 99; do
  NSFEM_LIB=build/knockouts/libnsfem_hip.so NSFEM_LEG_MAXOPS=$k scripts/r04_trace_script.sh r04_maxops_$k scripts/r04_legs_time.py 512 > /dev/null 2>&1
  echo "== maxops $k"; grep "k_mg_leg" gpurun_out/r04_maxops_$k/by_grid.csv | cut -d, -f1-6
done
