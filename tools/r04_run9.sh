#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r04i; mkdir -p $O
M=$R/ganspace_amd/lib_measure/libganspace_hip.so
GANSPACE_HIP_LIB=$M GS_TOPK_DEBUG=1 timeout 300 python tools/finalize_trace.py 10 2 exact 2>&1 | grep "chol_inv" | head -3
timeout 300 python tools/finalize_trace.py 100 3 both > $O/finalize.log 2>&1; grep "exact fin\|faithful" $O/finalize.log | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_topk.py -x -q > $O/t1.log 2>&1; echo "rc=$?"; grep -E "passed|failed|error|Error" $O/t1.log | tail -5
