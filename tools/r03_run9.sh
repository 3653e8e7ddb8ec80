#!/bin/bash
out=gpurun_out/r03k; mkdir -p $out
for rep in 1 2 3; do
  echo "== production (runtime-conditional phases)" >> $out/probe.log
  python tools/gram_probe.py 131072 512 f32 2>&1 | grep -E "gram_partial" | tail -1 >> $out/probe.log
  echo "== -DGS_WIDE_F32_UNCOND" >> $out/probe.log
  GANSPACE_HIP_LIB=ganspace_amd/lib_uncond/libganspace_hip.so python tools/gram_probe.py 131072 512 f32 2>&1 | grep -E "gram_partial" | tail -1 >> $out/probe.log
done
python tools/gram_probe.py 10000 512 f32 2>&1 | grep -E "gram_partial" | tail -1 >> $out/probe.log
python tools/gram_probe.py 131072 512 bf16x3 2>&1 | grep -E "gram_partial" | tail -1 >> $out/probe.log
cat $out/probe.log
