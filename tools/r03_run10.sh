#!/bin/bash
out=gpurun_out/r03l; mkdir -p $out
GS_GRAM_TRACE=1 GANSPACE_HIP_LIB=ganspace_amd/lib_trace/libganspace_hip.so python tools/gram_probe.py 524288 512 bf16 > $out/trace.log 2>&1
grep -v "Sampling" $out/trace.log | head -150
