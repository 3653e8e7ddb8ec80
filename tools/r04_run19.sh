#!/bin/bash
# round 4, call 20: solver read-backs through pinned scratch
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
timeout 300 python -m pytest tests/test_gpu_topk.py tests/test_gpu_whole_matrix.py -x -q 2>&1 | grep -E "passed|failed|^E  " | head
for v in 1 2 3; do
  timeout 300 python bench.py --no-extras --steps 20 --warmup 5 2> /dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.readlines()[-1]); print('prod', b['value'], b['ms_per_step'], b['breakdown']['update_loop_s'], b['breakdown']['finalize_eigensolve_s'], b['roofline']['in_job_avg_launch_us'])"
done
timeout 300 python tools/finalize_trace.py 100 3 both 2>&1 | grep "exact fin\|faithful" | cut -c1-200
