#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r04h; mkdir -p $O
tools/ubench/chol_leaf
timeout 300 python tools/finalize_trace.py 100 3 both > $O/finalize.log 2>&1; grep "exact fin\|faithful" $O/finalize.log | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_topk.py tests/test_gpu_merge.py tests/test_gpu_collective_shim.py tests/test_gpu_benchmarked_shapes.py -x -q > $O/t1.log 2>&1; echo "rc=$?"; grep -E "passed|failed|error|Error" $O/t1.log | tail -5
timeout 600 python bench.py --no-extras > $O/bench_ne.json 2> $O/bench_ne.err; python -c "
import json; d=json.load(open('$O/bench_ne.json')); print(d['value'], d['ms_per_step'], d['breakdown'], d['roofline']['frac'])"
timeout 300 python - <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
import bench
from ganspace_amd.estimators import IPCAEstimator
dev = torch.device('cuda', 0)
blocks = bench.make_blocks(100, dev)[0]
for rep in range(4):
    ef = IPCAEstimator(80, 'faithful'); ef.transformer._ensure(512)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for b in blocks: ef.fit_partial(b)
    ef.get_components(); torch.cuda.synchronize()
    print('faithful job (no per-block sync): %.2f ms = %.4f ms/block' % ((time.perf_counter()-t0)*1e3, (time.perf_counter()-t0)*10))
PY
