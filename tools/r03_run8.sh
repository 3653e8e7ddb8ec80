#!/bin/bash
out=gpurun_out/r03j; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "bf16" 2>&1 | tail -2 > $out/tests.log
echo "== production build: single-plane bf16, two k-steps in flight, unconditional phases" >> $out/probe.log
for rows in 524288 131072; do python tools/gram_probe.py $rows 512 bf16 2>&1 | grep -E "gram_partial" | tail -1 >> $out/probe.log; done
echo "== ablate build, mask 0: three fetch sets, runtime-conditional phases (the previous production form)" >> $out/probe.log
for rows in 524288 131072; do GANSPACE_HIP_LIB=ganspace_amd/lib_ablate/libganspace_hip.so python tools/gram_probe.py $rows 512 bf16 2>&1 | grep -E "gram_partial" | tail -1 >> $out/probe.log; done
cat $out/tests.log $out/probe.log
