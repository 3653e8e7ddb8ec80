#!/bin/bash
# round 5, call 2: sub-block skipping in rowgram_dma / tn_gemm, split-count sweep, cfg3 + cfg5 end to end, pinning probe
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r05b; mkdir -p $O
M=ganspace_amd/lib_measure/libganspace_hip.so
timeout 900 python -m pytest tests/test_gpu_benchmarked_shapes.py -x -q -k "32768 or panel or taller" > $O/t_shapes.log 2>&1; echo "shapes rc=$?"; grep -E "passed|failed|^E  " $O/t_shapes.log | head -12
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "smallside" > $O/t_parity.log 2>&1; echo "parity rc=$?"; grep -E "passed|failed|^E  " $O/t_parity.log | head -12
for cfg in "131072 f32" "32768 f32"; do set -- $cfg
  timeout 300 python tools/smallside_probe.py $1 2000 80 10 $2 2>&1 | grep block | tail -3
done | tee $O/ss_times.log
for ns in 5 15 20 30; do
  echo "nsplit=$ns"; GANSPACE_HIP_LIB=$M GS_SS_NSPLIT=$ns timeout 300 python tools/smallside_probe.py 131072 2000 80 8 f32 2>&1 | grep block | tail -2
done | tee $O/ss_nsplit.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ss131f -o s -- python tools/smallside_probe.py 131072 2000 80 10 f32 > /dev/null 2>&1
timeout 300 python tools/e2e_job.py cfg3 2> /dev/null | tail -1 | tee $O/e2e_cfg3.json
timeout 600 python tools/e2e_job.py cfg5 20000 250 2> $O/e2e_cfg5.err | tail -1 | tee $O/e2e_cfg5.json
timeout 600 python tools/e2e_job.py cfg5 100000 500 2> $O/e2e_cfg5b.err | tail -1 | tee $O/e2e_cfg5_n100k_b500.json
timeout 600 python -m pytest tests/test_gpu_decomposition.py -x -q > $O/t_dec.log 2>&1; echo "dec rc=$?"; grep -E "passed|failed|^E  " $O/t_dec.log | head -12
timeout 120 python tools/pin_probe.py 2>&1 | tee $O/pin_probe.log
python tools/summarize_r05.py $O 2>&1 | tee $O/summary.md | head -40
