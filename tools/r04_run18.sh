#!/bin/bash
# round 4, call 19: results of a fit in one device-to-host transfer
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r04u; mkdir -p $O
for v in 1 2 3; do
  timeout 300 python bench.py --no-extras --steps 20 --warmup 5 2> /dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.readlines()[-1]); print('prod', b['value'], b['ms_per_step'], b['breakdown']['update_loop_s'], b['breakdown']['finalize_eigensolve_s'], b['roofline']['in_job_avg_launch_us'])"
done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_merge.py -x -q 2>&1 | grep -E "passed|failed|^E  " | head
