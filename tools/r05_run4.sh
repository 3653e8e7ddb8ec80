#!/bin/bash
# round 5, call 4: tile order 8x8 vs 4x4, DMA issue spread over the MFMA groups, cfg5 end to end at n = 1e6
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r05d; mkdir -p $O
M=ganspace_amd/lib_measure/libganspace_hip.so
timeout 600 python -m pytest tests/test_gpu_benchmarked_shapes.py -x -q -k "32768 or panel" > $O/t_shapes.log 2>&1; echo "shapes rc=$?"; grep -E "passed|failed|^E  " $O/t_shapes.log | head
for v in "A_order8" "B_order4:GS_SS_ORDER_BLOCK=4" "C_spread:GS_ROWGRAM_SPREAD=1" "D_order17:GS_SS_ORDER_BLOCK=17"; do
  name=${v%%:*}; envs=""; [[ "$v" == *:* ]] && envs=${v#*:}
  echo "== $name $envs"
  env GANSPACE_HIP_LIB=$M $envs timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ss_$name -o s -- python tools/smallside_probe.py 131072 2000 80 8 f32 2>&1 | grep "block [67]" 
  python3 - <<PY
import csv,glob
f=glob.glob("$O/ss_$name/*kernel_stats.csv")
for r in csv.DictReader(open(f[0])):
    if "rowgram_dma" in r["Name"]: print("   rowgram_dma avg us", float(r["AverageNs"])/1e3, "calls", r["Calls"])
PY
done 2>&1 | tee $O/variants.log
GANSPACE_HIP_LIB=$M timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_order8 -o p -- python tools/smallside_probe.py 131072 2000 80 6 f32 > /dev/null 2>&1
GANSPACE_HIP_LIB=$M timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_l2_order8 -o p -- python tools/smallside_probe.py 131072 2000 80 6 f32 > /dev/null 2>&1
GANSPACE_HIP_LIB=$M GS_ROWGRAM_PROBE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ss_probe1 -o s -- python tools/smallside_probe.py 131072 2000 80 6 f32 > /dev/null 2>&1
timeout 600 python tools/e2e_job.py cfg5 1000000 500 2> $O/e2e_cfg5.err | tail -1 | tee $O/e2e_cfg5_n1e6.json
python tools/summarize_r05.py $O 2>&1 > $O/summary.md; grep -A3 "pmc_\|probe1" $O/summary.md | head -40
