#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
out=gpurun_out/r03x; mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_trace -o b -- python bench.py --no-extras > $out/bench_profiled.json 2> /dev/null
tail -1 $out/bench.err; ls $out/bench_trace | head
