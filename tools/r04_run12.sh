#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r04l; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_topk.py -x -q > $O/t1.log 2>&1; echo "rc=$?"; grep -E "passed|failed|^E  " $O/t1.log | head -20
timeout 300 python tools/finalize_trace.py 100 3 both > $O/finalize.log 2>&1; grep "exact fin\|faithful" $O/finalize.log | cut -c1-160
M=$R/ganspace_amd/lib_measure/libganspace_hip.so
for g in 32 48; do echo "guards $g"; GANSPACE_HIP_LIB=$M GS_SUBSPACE_EXTRA=$g timeout 300 python tools/finalize_trace.py 100 4 exact 2>&1 | grep "exact fin" | tail -2; done
echo "jacobi RR"; GANSPACE_HIP_LIB=$M GS_RR_JACOBI=1 timeout 300 python tools/finalize_trace.py 100 4 exact 2>&1 | grep "exact fin" | tail -2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ft -o f -- python tools/finalize_trace.py 100 6 exact > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r04l/ft/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:58]:58s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.2f} us")
PY
