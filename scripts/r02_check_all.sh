#!/bin/bash
# full GPU suite + smoke
cd /root/repo
mkdir -p gpurun_out/r02u
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r02u/gpu_tests.log 2>&1
echo "exit $?" >> gpurun_out/r02u/gpu_tests.log
tail -6 gpurun_out/r02u/gpu_tests.log
