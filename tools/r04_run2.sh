#!/bin/bash
# round 4, GPU call 2: launch floor; chol_inv v2 (branch-free leaf, MFMA panel / trailing)
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r04b; mkdir -p $O
tools/ubench/launch_floor > $O/launch_floor.log 2>&1; cat $O/launch_floor.log
timeout 600 python -m pytest tests/test_gpu_topk.py -x -q > $O/topk_tests.log 2>&1; echo "topk tests rc=$?"
tail -5 $O/topk_tests.log
timeout 300 python tools/finalize_trace.py 100 4 both 2>&1 | grep "exact\|faithful" | cut -c1-300 > $O/finalize.log; cat $O/finalize.log
M=$R/ganspace_amd/lib_measure/libganspace_hip.so
GANSPACE_HIP_LIB=$M GS_TOPK_DEBUG=1 timeout 300 python tools/finalize_trace.py 10 2 exact 2>&1 | grep "chol_inv\|jacobi\|exact" | head -12 > $O/finalize_debug.log; cat $O/finalize_debug.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/finalize_trace -o f -- python tools/finalize_trace.py 100 6 both > $O/finalize_profiled.log 2> /dev/null
find $O/finalize_trace -name "*kernel_stats.csv" -exec cp {} $O/finalize_kernel_stats.csv \;
cut -d, -f1-4 $O/finalize_kernel_stats.csv | cut -c1-150 | head -12
timeout 600 python -m pytest tests/test_gpu_benchmarked_shapes.py -x -q 2>&1 | tail -3
