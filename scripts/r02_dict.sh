#!/bin/bash
cd /root/repo
O=gpurun_out/r02d; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -x -q -k "block_dictionaries" > $O/tests.log 2>&1; tail -30 $O/tests.log
