#!/bin/bash
# round 5, call 3: device z generator, convolution through the HIP GEMM, LEAN mm64 in production, FETCH/WRITE passes
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r05c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_zgen.py -x -q > $O/t_zgen.log 2>&1; echo "zgen rc=$?"; grep -E "passed|failed|^E  " $O/t_zgen.log | head -12
timeout 300 python tools/zgen_device_probe.py 2>&1 | grep -E "zgen|make_blocks" | tee $O/zgen_probe.log
timeout 900 python -m pytest tests/test_gpu_decomposition.py tests/test_gpu_topk.py -x -q > $O/t_dec.log 2>&1; echo "dec+topk rc=$?"; grep -E "passed|failed|^E  " $O/t_dec.log | head -12
timeout 300 python tools/e2e_job.py cfg5 20000 250 2> $O/e2e_cfg5.err | tail -1 | tee $O/e2e_cfg5.json
timeout 600 python tools/e2e_job.py cfg5 100000 500 2> $O/e2e_cfg5b.err | tail -1 | tee $O/e2e_cfg5_n100k_b500.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/e2e5 -o e -- python tools/e2e_job.py cfg5 20000 250 > /dev/null 2>&1
for i in 1 2; do timeout 300 python tools/e2e_job.py cfg3 2> /dev/null | tail -1; done | tee $O/e2e_cfg3.json
timeout 300 python tools/e2e_cfg2.py 2>&1 | grep -E "E2E|cos" | tee $O/e2e_cfg2.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python tools/smallside_probe.py 131072 2000 80 6 f32 > /dev/null 2>&1
done
timeout 300 python bench.py --no-extras --steps 20 --warmup 5 2> /dev/null | tail -1 > $O/bench_noextras.json; python3 -c "
import json; b=json.load(open('$O/bench_noextras.json')); print('bench', b['value'], b['ms_per_step'], b['roofline']['frac'], b.get('breakdown'))"
python tools/summarize_r05.py $O 2>&1 | tee $O/summary.md | head -60
