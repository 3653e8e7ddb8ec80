#!/bin/bash
out=gpurun_out/r03r; mkdir -p $out
python tools/data_probe.py > $out/data_probe.log 2>&1
cat $out/data_probe.log | tail -32
