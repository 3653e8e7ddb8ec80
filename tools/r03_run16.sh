#!/bin/bash
out=gpurun_out/r03t; mkdir -p $out
python -m pytest tests/test_gpu_decomposition.py tests/test_gpu_distributed.py -q -m gpu -k "z_space or cfg3 or cfg5_conv_features_small or z_exact" 2>&1 | tail -5 > $out/tests.log
python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/bench.err
cat $out/tests.log; tail -3 $out/bench.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03t/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['warmup'], d['roofline']['avg_launch_us'], d['roofline']['frac'])
print(d.get('regression_cfg4_share'))
print(d['roofline_hbm'])
PY
