#!/bin/bash
# GPU run 4 of round 3: plain-bf16 contraction (tests, timing, ablations), Jacobi LP=16 as default, z-gen probe, bench
out=gpurun_out/r03d; mkdir -p $out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_topk.py tests/test_gpu_merge.py -x -q -m gpu > $out/tests.log 2>&1; echo "tests rc=$?" >> $out/tests.log
for prec in bf16 bf16x3 f32; do python tools/gram_probe.py 131072 512 $prec 2>&1 | grep -E "gram_partial|update" | tail -2 >> $out/probe.log; done
python tools/gram_probe.py 50000 512 bf16 2>&1 | grep gram_partial | tail -1 >> $out/probe.log
python tools/gram_probe.py 10000 512 bf16 2>&1 | grep gram_partial | tail -1 >> $out/probe.log
for ab in 0 1 2 4 3 6 5; do
  echo "== bf16 wide, ablate mask $ab (1 no MFMA, 2 no split/LDS write, 4 no loads)" >> $out/probe.log
  GANSPACE_HIP_LIB=ganspace_amd/lib_ablate/libganspace_hip.so GS_GRAM_ABLATE=$ab python tools/gram_probe.py 131072 512 bf16 2>&1 | grep gram_partial | tail -1 >> $out/probe.log
done
python tools/zgen_probe.py 2>&1 | grep -v Sampling >> $out/probe.log
python tools/finalize_trace.py 100 5 both 2>&1 | grep -v Sampling | cut -c1-400 >> $out/probe.log
python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/bench.err
tail -4 $out/tests.log; cat $out/probe.log
