#!/bin/bash
# round 3, final run A: the whole GPU suite + smoke + the default bench
out=gpurun_out/r03q; mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/smoke.log
python -m pytest tests -q -m gpu > $out/tests.log 2>&1; echo "tests rc=$?" >> $out/tests.log
python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/bench.err
tail -2 $out/smoke.log; tail -6 $out/tests.log; tail -2 $out/bench.err | cut -c1-300
