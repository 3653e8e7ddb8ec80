#!/bin/bash
out=gpurun_out/r03aa; mkdir -p $out
timeout 55 python bench.py --no-extras > $out/bench.json 2> $out/bench.err; echo "rc=$?"
python -c "
import json; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['breakdown'])"
grep -i "resident" $out/bench.err | tail -1
