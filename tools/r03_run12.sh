#!/bin/bash
out=gpurun_out/r03n; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "bf16" 2>&1 | tail -3 > $out/tests.log
echo "== LDS-DMA kernel, opposite phase order in the two waves of a SIMD" >> $out/probe.log
for rows in 524288 131072; do python tools/gram_probe.py $rows 512 bf16 2>&1 | grep -E "gram_partial" | tail -1 >> $out/probe.log; done
echo "== LDS-DMA kernel, same order (GS_BF16_SAME_ORDER=1)" >> $out/probe.log
for rows in 524288 131072; do GS_BF16_SAME_ORDER=1 python tools/gram_probe.py $rows 512 bf16 2>&1 | grep -E "gram_partial" | tail -1 >> $out/probe.log; done
GS_GRAM_TRACE=1 GANSPACE_HIP_LIB=ganspace_amd/lib_trace/libganspace_hip.so python tools/gram_probe.py 524288 512 bf16 2>&1 | grep -A15 "workgroup 0 wave" >> $out/probe.log
cat $out/tests.log $out/probe.log
