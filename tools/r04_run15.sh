#!/bin/bash
# round 4, call 15: 4 x 8 block tridiagonalisation, shift from the first 8192 rows, kernel-attached ev_comp (measure knob)
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r04q; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_topk.py -x -q > $O/t1.log 2>&1; echo "topk rc=$?"; grep -E "passed|failed|^E  " $O/t1.log | head -20
timeout 300 python tools/finalize_trace.py 100 3 both > $O/finalize.log 2>&1; grep "exact fin\|faithful" $O/finalize.log | cut -c1-160
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ft -o f -- python tools/finalize_trace.py 100 6 exact > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r04q/ft/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:58]:58s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.2f} us")
PY
M=$R/ganspace_amd/lib_measure/libganspace_hip.so
for v in base ext base ext; do
  if [ $v = ext ]; then export GS_GRAM_EXT_EVENT=1; else unset GS_GRAM_EXT_EVENT; fi
  GANSPACE_HIP_LIB=$M timeout 300 python bench.py --no-extras --steps 20 --warmup 5 2> /dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.readlines()[-1]); print('$v', b['value'], b['ms_per_step'], b['breakdown']['update_loop_s'], b['breakdown']['finalize_eigensolve_s'], b['roofline']['in_job_avg_launch_us'])"
done
unset GS_GRAM_EXT_EVENT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|^E  " | head
