#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r04d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_topk.py tests/test_gpu_merge.py -x -q > $O/t1.log 2>&1; echo "rc=$?"; grep -E "passed|failed|error" $O/t1.log | tail -3
M=$R/ganspace_amd/lib_measure/libganspace_hip.so
GANSPACE_HIP_LIB=$M GS_TOPK_DEBUG=1 timeout 300 python tools/finalize_trace.py 30 1 faithful > $O/dbg.log 2>&1; grep "invsub" $O/dbg.log | tail -8
for g in 32 48; do echo "guards $g"; GANSPACE_HIP_LIB=$M GS_SUBSPACE_EXTRA=$g timeout 300 python tools/finalize_trace.py 100 4 exact > $O/g$g.log 2>&1; grep "exact fin" $O/g$g.log | tail -2; tail -3 $O/g$g.log | cut -c1-200; done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $O/t2.log 2>&1; echo "rc=$?"; grep -E "passed|failed|error" $O/t2.log | tail -3
timeout 900 python -m pytest tests/test_gpu_whole_matrix.py -x -q > $O/t3.log 2>&1; echo "rc=$?"; grep -E "passed|failed|error" $O/t3.log | tail -3
