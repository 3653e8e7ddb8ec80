#!/bin/bash
# round 5, call 8 (profiles only, final source): kernel trace of the cfg3 end-to-end job, the driver's bench flags, SQ counters
# of the small side's launches with the sub-block skipping in
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r05h; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/e2e3 -o e -- python tools/e2e_job.py cfg3 > $O/e2e_cfg3_profiled.json 2> /dev/null
timeout 300 python bench.py --steps 20 --warmup 5 --no-wide --no-e2e 2> /dev/null | tail -1 > $O/bench_driver_flags.json
python3 -c "
import json; b=json.load(open('$O/bench_driver_flags.json')); print('bench(driver flags)', b['value'], b['ms_per_step'], b['roofline']['frac'], b['breakdown'])"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o p -- python tools/smallside_probe.py 131072 2000 80 6 f32 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq32 -o p -- python tools/smallside_probe.py 32768 2000 80 6 f32 > /dev/null 2>&1
python tools/summarize_r05.py $O 2>&1 | tee $O/summary.md | head -60
