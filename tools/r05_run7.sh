#!/bin/bash
# round 5, call 7: the small side's invariant-subspace verdict read one block late (smallside_resolve)
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r05g; mkdir -p $O
M=ganspace_amd/lib_measure/libganspace_hip.so
timeout 900 python -m pytest tests/test_gpu_benchmarked_shapes.py tests/test_gpu_merge.py -x -q > $O/t_shapes.log 2>&1; echo "shapes+merge rc=$?"; grep -E "passed|failed|^E  " $O/t_shapes.log | head
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "smallside or deferred or wide" > $O/t_parity.log 2>&1; echo "parity rc=$?"; grep -E "passed|failed|^E  " $O/t_parity.log | head
timeout 900 python -m pytest tests/test_gpu_decomposition.py tests/test_gpu_distributed.py tests/test_gpu_whole_matrix.py -x -q > $O/t_dec.log 2>&1; echo "dec+dist+whole rc=$?"; grep -E "passed|failed|^E  " $O/t_dec.log | head
for i in 1 2; do GANSPACE_HIP_LIB=$M timeout 300 python tools/e2e_job.py cfg3 2> /dev/null | tail -1; done | tee $O/e2e_cfg3_deferred.json
for i in 1 2; do GANSPACE_HIP_LIB=$M GS_SS_SYNC_VERDICT=1 timeout 300 python tools/e2e_job.py cfg3 2> /dev/null | tail -1; done | tee $O/e2e_cfg3_sync.json
timeout 300 python tools/smallside_probe.py 32768 2000 80 10 f32 2>&1 | grep block | tail -3 | tee $O/ss32.log
timeout 300 python tools/e2e_job.py cfg5 100000 500 2> /dev/null | tail -1 | tee $O/e2e_cfg5_n100k.json
