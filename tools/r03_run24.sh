#!/bin/bash
out=gpurun_out/r03ac; mkdir -p $out
timeout 40 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "second_stream_fold" > $out/tests.log 2>&1
grep -E "passed|failed|error" $out/tests.log | tail -2
