#!/bin/bash
# round 4 profile collection (run ON THE GPU BOX through gpurun).  rocprofv3 kernel-trace stats and PMC passes are
# separate runs (never --pmc together with sys / hip / hsa tracing).
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r04p; mkdir -p $O
tools/ubench/launch_floor > $O/launch_floor.log 2>&1
tools/ubench/chol_leaf > $O/chol_leaf.log 2>&1
for r in 1000000 131072; do python tools/gram_probe.py $r 512 bf16 2>&1 | grep gram_partial | tail -1 >> $O/gram_probe.log; done
python tools/gram_probe.py 131072 512 f32 2>&1 | grep gram_partial | tail -1 >> $O/gram_probe.log
# 1. kernel stats: headline job (+ modes), finalize / faithful chains, small side at d = 131 072 and 32 768
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -o b -- python bench.py --no-cpu-baseline --no-wide --no-e2e > $O/bench_profiled.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/finalize_trace -o f -- python tools/finalize_trace.py 100 6 both > $O/finalize_trace.log 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ss131f_trace -o s -- python tools/smallside_probe.py 131072 2000 80 12 f32 > $O/ss131_f32.log 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ss32_trace -o s -- python tools/smallside_probe.py 32768 2000 80 12 f32 > $O/ss32_f32.log 2> /dev/null
# 2. PMC: HBM traffic of the Gram kernels (f32 wide, bf16 wide)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_f32_$c -o p -- python tools/gram_probe.py 131072 512 f32 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_bf16_$c -o p -- python tools/gram_probe.py 1000000 512 bf16 > /dev/null 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_f32_sq -o p -- python tools/gram_probe.py 131072 512 f32 > /dev/null 2>&1
python tools/summarize_r04.py $O > $O/summary.md 2> $O/summary.err
head -120 $O/summary.md
python tools/job_timeline.py $O/bench_trace > $O/job_timeline.md 2>&1; tail -25 $O/job_timeline.md
