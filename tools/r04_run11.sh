#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r04k; mkdir -p $O
M=$R/ganspace_amd/lib_measure/libganspace_hip.so
for a in 0 1 2 4 7; do
  GANSPACE_HIP_LIB=$M GS_CHOL_ABLATE=$a timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$a -o f -- python tools/finalize_trace.py 10 3 exact > /dev/null 2>&1
  f=$(find $O/t$a -name "*kernel_stats.csv" | head -1)
  echo "ablate $a: $(grep chol_inv $f | cut -d, -f2-4,6)"
done
