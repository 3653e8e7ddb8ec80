#!/bin/bash
# round 5, call 1: the LDS-DMA rowgram kernel / panel-blocked M / project_rows - parity first, then times, traces, PMC
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r05a; mkdir -p $O
M=ganspace_amd/lib_measure/libganspace_hip.so
timeout 900 python -m pytest tests/test_gpu_benchmarked_shapes.py -x -q > $O/t_shapes.log 2>&1; echo "shapes rc=$?"; grep -E "passed|failed|^E  " $O/t_shapes.log | head -12
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "smallside or project_rows or linear" > $O/t_parity.log 2>&1; echo "parity rc=$?"; grep -E "passed|failed|^E  " $O/t_parity.log | head -12
timeout 600 python -m pytest tests/test_gpu_decomposition.py -x -q -k "cfg3 or z_space or regression" > $O/t_dec.log 2>&1; echo "dec rc=$?"; grep -E "passed|failed|^E  " $O/t_dec.log | head -12
for cfg in "131072 f32" "131072 bf16x6" "32768 f32"; do set -- $cfg
  timeout 300 python tools/smallside_probe.py $1 2000 80 10 $2 2>&1 | grep block | tail -4
done | tee $O/ss_times.log
# kernel traces
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ss131f -o s -- python tools/smallside_probe.py 131072 2000 80 10 f32 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ss32f -o s -- python tools/smallside_probe.py 32768 2000 80 10 f32 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ss131b -o s -- python tools/smallside_probe.py 131072 2000 80 10 bf16x6 > /dev/null 2>&1
for p in 1 2; do
  GANSPACE_HIP_LIB=$M GS_ROWGRAM_PROBE=$p timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ss131f_probe$p -o s -- python tools/smallside_probe.py 131072 2000 80 6 f32 > /dev/null 2>&1
done
# PMC (separate passes, kernel-trace only)
timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_mem -o p -- python tools/smallside_probe.py 131072 2000 80 6 f32 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o p -- python tools/smallside_probe.py 131072 2000 80 6 f32 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_l2 -o p -- python tools/smallside_probe.py 131072 2000 80 6 f32 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_lds -o p -- python tools/smallside_probe.py 131072 2000 80 6 f32 > /dev/null 2>&1
# cfg3 end to end: phases + kernel trace
timeout 300 python tools/e2e_job.py cfg3 2> /dev/null | tail -1 | tee $O/e2e_cfg3.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/e2e3 -o e -- python tools/e2e_job.py cfg3 > $O/e2e_cfg3_profiled.json 2> /dev/null
# MM64_LEAN: validate, time
GANSPACE_HIP_LIB=$M GS_MM64_LEAN=1 timeout 600 python -m pytest tests/test_gpu_topk.py -x -q > $O/t_lean.log 2>&1; echo "lean rc=$?"; grep -E "passed|failed|^E  " $O/t_lean.log | head
GANSPACE_HIP_LIB=$M timeout 200 python tools/finalize_trace.py 100 3 exact 2>&1 | grep "exact fin" | cut -c1-160 | tee $O/fin_base.log
GANSPACE_HIP_LIB=$M GS_MM64_LEAN=1 timeout 200 python tools/finalize_trace.py 100 3 exact 2>&1 | grep "exact fin" | cut -c1-160 | tee $O/fin_lean.log
python tools/summarize_r05.py $O 2>&1 | tee $O/summary.md | head -150
