#!/bin/bash
out=gpurun_out/r03s; mkdir -p $out
for w in 5 5 50 200 1000; do
  python bench.py --no-extras --warmup $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('warmup',d['warmup'],'value %.4g'%d['value'],'ms/step',d['ms_per_step'],'launch_us',d['roofline']['avg_launch_us'],d['breakdown'])" >> $out/warm.log
done
cat $out/warm.log
