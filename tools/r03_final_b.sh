#!/bin/bash
out=gpurun_out/r03i; mkdir -p $out
for i in 1 2 3 4; do python -m pytest tests/test_gpu_parity.py -q -m gpu -k "bf16" 2>&1 | tail -2 >> $out/bf16_tests.log; done
python tools/gram_probe.py 524288 512 bf16 2>&1 | grep gram_partial | tail -1 >> $out/bf16_tests.log
python tools/gram_probe.py 131072 512 bf16 2>&1 | grep gram_partial | tail -1 >> $out/bf16_tests.log
cat $out/bf16_tests.log
bash tools/r03_profiles.sh
