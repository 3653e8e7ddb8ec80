#!/bin/bash
# round 5, call 5: prefetch of the next block on a second stream, random_stdevs on the device, 4-wave z generator
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r05e; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_zgen.py -x -q > $O/t_zgen.log 2>&1; echo "zgen rc=$?"; grep -E "passed|failed|^E  " $O/t_zgen.log | head
timeout 200 python tools/zgen_device_probe.py 2>&1 | grep -E "zgen|make_blocks" | tee $O/zgen_probe.log
timeout 900 python -m pytest tests/test_gpu_decomposition.py tests/test_gpu_distributed.py -x -q > $O/t_dec.log 2>&1; echo "dec+dist rc=$?"; grep -E "passed|failed|^E  " $O/t_dec.log | head -12
for i in 1 2; do timeout 300 python tools/e2e_job.py cfg3 2> /dev/null | tail -1; done | tee $O/e2e_cfg3.json
GANSPACE_NO_PREFETCH=1 timeout 300 python tools/e2e_job.py cfg3 2> /dev/null | tail -1 | tee $O/e2e_cfg3_noprefetch.json
timeout 300 python tools/e2e_job.py cfg2 2> /dev/null | tail -1 | tee $O/e2e_cfg2.json
timeout 300 python tools/e2e_job.py cfg2f 2> /dev/null | tail -1 | tee $O/e2e_cfg2f.json
timeout 300 python tools/e2e_job.py cfg5 100000 500 2> $O/e2e_cfg5.err | tail -1 | tee $O/e2e_cfg5_n100k.json
