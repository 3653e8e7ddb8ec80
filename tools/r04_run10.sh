#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r04j; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/finalize_trace -o f -- python tools/finalize_trace.py 100 6 both > $O/finalize_profiled.log 2> /dev/null
find $O/finalize_trace -name "*kernel_stats.csv" -exec cp {} $O/finalize_kernel_stats.csv \;
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r04j/finalize_kernel_stats.csv')))
for r in rows[:16]:
    print(f"{r['Name'][:58]:58s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.2f} us  min {int(r['MinNs'])/1e3:.2f}")
PY
grep "exact fin\|faithful" $O/finalize_profiled.log | cut -c1-160 | tail -4
