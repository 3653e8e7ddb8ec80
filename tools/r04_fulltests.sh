#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r04_tests; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "rc=$?"; tail -5 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
