#!/bin/bash
out=gpurun_out/r03e; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bf16" > $out/tests.log 2>&1; echo "tests rc=$?" >> $out/tests.log
tools/ubench/read_bw > $out/read_bw.log 2>&1
for rows in 524288 131072; do python tools/gram_probe.py $rows 512 bf16 2>&1 | grep -E "gram_partial" | tail -1 >> $out/probe.log; done
for ab in 0 3 6 5 4; do
  echo "== bf16 wide 524288 rows, ablate mask $ab (1 no MFMA, 2 no split/LDS write, 4 no loads)" >> $out/probe.log
  GANSPACE_HIP_LIB=ganspace_amd/lib_ablate/libganspace_hip.so GS_GRAM_ABLATE=$ab python tools/gram_probe.py 524288 512 bf16 2>&1 | grep gram_partial | tail -1 >> $out/probe.log
done
python bench.py --no-wide > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/bench.err
tail -3 $out/tests.log; cat $out/read_bw.log $out/probe.log
