#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02s
timeout -k 10 1100 python -m pytest tests/test_gpu_partition.py -x -q > gpurun_out/r02s/test.log 2>&1
echo "exit $?" >> gpurun_out/r02s/test.log
tail -15 gpurun_out/r02s/test.log
