#!/bin/bash
# round 4, GPU call 3: no ring memset, second-order Loewdin pass, per-device attributes, tn_gemm without scratch
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r04c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_topk.py tests/test_gpu_merge.py -x -q 2>&1 | tail -3
timeout 300 python tools/finalize_trace.py 100 4 both 2>&1 | grep "exact\|faithful" | cut -c1-250 > $O/finalize.log; cat $O/finalize.log
M=$R/ganspace_amd/lib_measure/libganspace_hip.so
for g in 32 48; do echo "guards $g"; GANSPACE_HIP_LIB=$M GS_SUBSPACE_EXTRA=$g timeout 300 python tools/finalize_trace.py 100 4 exact 2>&1 | grep "exact" | tail -2; done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
timeout 300 python tools/smallside_probe.py 131072 2000 80 10 f32 2>&1 | tail -2 | cut -c1-200
timeout 300 python tools/smallside_probe.py 32768 2000 80 10 f32 2>&1 | tail -2 | cut -c1-200
