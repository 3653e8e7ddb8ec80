#!/bin/bash
out=gpurun_out/r03o; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "bf16" 2>&1 | tail -3 > $out/tests.log
for rows in 1048576 1000000 524288; do python tools/gram_probe.py $rows 512 bf16 2>&1 | grep -E "gram_partial" | tail -1 >> $out/probe.log; done
cat $out/tests.log $out/probe.log
