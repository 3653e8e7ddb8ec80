#!/bin/bash
# round 5, call 10: convolution GEMMs from panel-blocked operands (gs_gemm_blocked_nt) - parity, then cfg5 end to end A/B
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r05j; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "blocked or project_rows or linear" > $O/t_blocked.log 2>&1; echo "blocked rc=$?"; grep -E "passed|failed|^E  " $O/t_blocked.log | head
timeout 300 python tools/e2e_job.py cfg5 100000 500 2> /dev/null | tail -1 | tee $O/e2e_cfg5_blocked.json
GANSPACE_CONV=strided timeout 300 python tools/e2e_job.py cfg5 100000 500 2> /dev/null | tail -1 | tee $O/e2e_cfg5_strided.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/e2e5 -o e -- python tools/e2e_job.py cfg5 20000 500 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r05j/e2e5/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print(f"{r['Name'].split('(')[0][-60:]:60s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f} %")
PY
timeout 900 python -m pytest tests/test_gpu_decomposition.py -x -q > $O/t_dec.log 2>&1; echo "dec rc=$?"; grep -E "passed|failed|^E  " $O/t_dec.log | head
