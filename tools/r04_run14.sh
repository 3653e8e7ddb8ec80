#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x > gpurun_out/r04_run14.log 2>&1
echo "rc=$?" >> gpurun_out/r04_run14.log
tail -60 gpurun_out/r04_run14.log
