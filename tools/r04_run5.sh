#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r04e; mkdir -p $O
timeout 300 python tools/finalize_trace.py 100 3 both > $O/finalize.log 2>&1; grep "exact fin\|faithful" $O/finalize.log | cut -c1-260
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_topk.py tests/test_gpu_merge.py tests/test_gpu_collective_shim.py -x -q > $O/t1.log 2>&1; echo "rc=$?"; grep -E "passed|failed|error|Error" $O/t1.log | tail -5
timeout 900 python -m pytest tests/test_gpu_decomposition.py -x -q -k "not cfg2_full" > $O/t2.log 2>&1; echo "rc=$?"; grep -E "passed|failed|error|Error" $O/t2.log | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
