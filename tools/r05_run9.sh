#!/bin/bash
# round 5, call 9: start one workgroup of each co-resident pair late (GS_ROWGRAM_STAGGER x 1024 clk) - does it break the lockstep?
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r05i; mkdir -p $O
M=ganspace_amd/lib_measure/libganspace_hip.so
for st in 0 2 4 7; do
  echo "== stagger $st"
  GANSPACE_HIP_LIB=$M GS_ROWGRAM_STAGGER=$st timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ss_st$st -o s -- python tools/smallside_probe.py 131072 2000 80 8 f32 2>&1 | grep "block [67]"
  python3 - <<PY
import csv,glob
f=glob.glob("$O/ss_st$st/*kernel_stats.csv")
for r in csv.DictReader(open(f[0])):
    if "rowgram_dma" in r["Name"]: print("   rowgram_dma avg us", float(r["AverageNs"])/1e3, "calls", r["Calls"])
PY
done 2>&1 | tee $O/stagger.log
for st in 0 4; do
  GANSPACE_HIP_LIB=$M GS_ROWGRAM_STAGGER=$st timeout 300 python tools/smallside_probe.py 32768 2000 80 10 f32 2>&1 | grep "block [789]" | sed "s/^/st$st /"
done | tee -a $O/stagger.log
