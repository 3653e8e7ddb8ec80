#!/bin/bash
out=gpurun_out/r03f; mkdir -p $out
for prec in f32 bf16x3 bf16; do python tools/gram_probe.py 131072 512 $prec 2>&1 | grep -E "gram_partial" | tail -1 >> $out/probe.log; done
python tools/gram_probe.py 524288 512 bf16 2>&1 | grep -E "gram_partial" | tail -1 >> $out/probe.log
for prec in f32 bf16x6 bf16x3 bf16; do python tools/gram_probe.py 10000 512 $prec 2>&1 | grep -E "gram_partial" | tail -1 >> $out/probe.log; done
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gram or exact_mode or bf16" > $out/tests.log 2>&1; echo "tests rc=$?" >> $out/tests.log
tail -3 $out/tests.log; cat $out/probe.log
