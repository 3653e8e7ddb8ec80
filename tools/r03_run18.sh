#!/bin/bash
out=gpurun_out/r03v; mkdir -p $out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_benchmarked_shapes.py tests/test_gpu_merge.py -q -m gpu -x -k "not smallside and not faithful and not eigh and not mapping and not linear" > $out/tests.log 2>&1
grep -E "passed|failed|error" $out/tests.log | tail -3
grep -E "^FAILED|^ERROR|Error" $out/tests.log | head -10
