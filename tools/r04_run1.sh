#!/bin/bash
# round 4, GPU call 1: new float64 building blocks (chol_inv, mm64) - unit tests, chain timings, kernel stats, A/B
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r04a; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_topk.py -x -q > $O/topk_tests.log 2>&1; echo "topk tests rc=$?" 
tail -5 $O/topk_tests.log
timeout 300 python tools/finalize_trace.py 100 6 both > $O/finalize.log 2>&1; echo "finalize rc=$?"
cut -c1-400 $O/finalize.log
M=$R/ganspace_amd/lib_measure/libganspace_hip.so
GANSPACE_HIP_LIB=$M GS_CHOL_R3=1 timeout 300 python tools/finalize_trace.py 100 4 both 2>&1 | cut -c1-300 > $O/finalize_cholr3.log
GANSPACE_HIP_LIB=$M GS_GEMM_VALU=1 timeout 300 python tools/finalize_trace.py 100 4 both 2>&1 | cut -c1-300 > $O/finalize_valu.log
GANSPACE_HIP_LIB=$M GS_TOPK_DEBUG=1 timeout 300 python tools/finalize_trace.py 10 2 exact 2>&1 | grep "chol_inv\|jacobi\|exact" | head -40 > $O/finalize_debug.log
cat $O/finalize_cholr3.log $O/finalize_valu.log; cat $O/finalize_debug.log | head -30
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/finalize_trace -o f -- python tools/finalize_trace.py 100 6 both > $O/finalize_profiled.log 2> /dev/null
cp $O/finalize_trace/*/f_kernel_stats.csv $O/finalize_kernel_stats.csv 2>/dev/null || find $O/finalize_trace -name "*kernel_stats.csv" -exec cp {} $O/finalize_kernel_stats.csv \;
cut -d, -f1-4 $O/finalize_kernel_stats.csv | cut -c1-150 | head -40
timeout 300 python tools/smallside_probe.py 32768 2000 80 10 f32 > $O/ss32.log 2>&1; tail -3 $O/ss32.log | cut -c1-400
timeout 300 python tools/smallside_probe.py 131072 2000 80 10 f32 > $O/ss131.log 2>&1; tail -3 $O/ss131.log | cut -c1-400
