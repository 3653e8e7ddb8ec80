#!/bin/bash
# round 5, last call: the whole GPU suite + smoke on the final source (blocked convolution path in), cfg5 / cfg3 end to end
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r05k; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 300 python tools/e2e_job.py cfg5 1000000 500 2> /dev/null | tail -1 | tee $O/e2e_cfg5_n1e6.json
timeout 300 python tools/e2e_job.py cfg3 2> /dev/null | tail -1 | tee $O/e2e_cfg3.json
