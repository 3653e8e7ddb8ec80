#!/bin/bash
# GPU run 3 of round 3: full GPU suite (merge, whole-matrix, distributed, small-side fusions) + A/B probes
out=gpurun_out/r03c; mkdir -p $out
python -m pytest tests -x -q -m gpu > $out/tests.log 2>&1; echo "tests rc=$?" >> $out/tests.log
for lp in 8 16; do
  echo "== jacobi LP=$lp" >> $out/probe.log
  GS_JACOBI_LP=$lp GS_TOPK_DEBUG=1 python tools/finalize_trace.py 100 4 exact 2>&1 | grep -v Sampling | grep -E "exact finalize|jacobi p=" | tail -8 >> $out/probe.log
done
for wg in 160 640 1280; do
  echo "== gemm target wgs=$wg (small side d=131072, bf16x6 then f32)" >> $out/probe.log
  GS_GEMM_TARGET_WGS=$wg python tools/smallside_probe.py 131072 2000 80 10 bf16x6 2>&1 | tail -3 >> $out/probe.log
  GS_GEMM_TARGET_WGS=$wg python tools/smallside_probe.py 131072 2000 80 10 f32 2>&1 | tail -2 >> $out/probe.log
done
echo "== small side d=32768 f32" >> $out/probe.log
python tools/smallside_probe.py 32768 2000 80 10 f32 2>&1 | tail -2 >> $out/probe.log
python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/bench.err
tail -6 $out/tests.log; cat $out/probe.log
