#!/bin/bash
# round 4, calls 17-18: tridiag_reduce with compile-time skips and an LDS-only barrier; chol_inv loads in one round trip
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r04t; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_topk.py -x -q > $O/t1.log 2>&1; echo "topk rc=$?"; grep -E "passed|failed|^E  " $O/t1.log | head -20
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ft -o f -- python tools/finalize_trace.py 100 6 exact > $O/ft.log 2>&1
grep "exact fin" $O/ft.log | cut -c1-100
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r04t/ft/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{r['Name'][:58]:58s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.2f} us")
PY
timeout 300 python tools/finalize_trace.py 100 3 faithful 2>&1 | grep "faithful" | cut -c1-200
