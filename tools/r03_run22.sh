#!/bin/bash
out=gpurun_out/r03z; mkdir -p $out
timeout 115 python -m pytest tests/test_gpu_distributed.py -q -m gpu -x -k "faithful" > $out/tests.log 2>&1
grep -E "passed|failed|error" $out/tests.log | tail -3
