#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02s
timeout -k 10 900 python -m pytest tests/test_gpu_partition.py -x -q -k "rcb_partitioned" -s > gpurun_out/r02s/test.log 2>&1
echo "exit $?" >> gpurun_out/r02s/test.log
tail -30 gpurun_out/r02s/test.log
