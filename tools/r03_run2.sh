#!/bin/bash
# GPU run 2 of round 3: full GPU suite + graph / guard-column experiments on the exact finalize + bench
out=gpurun_out/r03b; mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/smoke.log
python -m pytest tests -x -q -m gpu > $out/tests.log 2>&1; echo "tests rc=$?" >> $out/tests.log
for extra in default 16 32; do
  for ng in 0 1; do
    envs=""; [ "$extra" != default ] && envs="GS_SUBSPACE_EXTRA=$extra"; [ $ng = 1 ] && envs="$envs GS_NO_GRAPHS=1"
    echo "== extra=$extra nographs=$ng" >> $out/finalize.log
    env $envs python tools/finalize_trace.py 100 6 both >> $out/finalize.log 2>&1
  done
done
python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/bench.err
tail -3 $out/smoke.log; tail -5 $out/tests.log; cat $out/finalize.log | grep -v "Sampling" | tail -60
