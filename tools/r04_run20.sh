#!/bin/bash
# round 4, call 22: the integration-level GPU test files once more on the final source
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r04w; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_decomposition.py tests/test_gpu_distributed.py -q --durations=12 > $O/t.log 2>&1; echo "rc=$?"
grep -E "passed|failed|^E  |^FAILED|s call|s setup" $O/t.log | head -30
