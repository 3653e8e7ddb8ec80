#!/bin/bash
out=gpurun_out/r03m; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "bf16" 2>&1 | tail -4 > $out/tests.log
echo "== LDS-DMA kernel" >> $out/probe.log
for rows in 524288 131072 20000; do python tools/gram_probe.py $rows 512 bf16 2>&1 | grep -E "gram_partial" | tail -1 >> $out/probe.log; done
echo "== register-staged kernel (GS_BF16_NO_GLDS=1)" >> $out/probe.log
for rows in 524288 131072; do GS_BF16_NO_GLDS=1 python tools/gram_probe.py $rows 512 bf16 2>&1 | grep -E "gram_partial" | tail -1 >> $out/probe.log; done
cat $out/tests.log $out/probe.log
