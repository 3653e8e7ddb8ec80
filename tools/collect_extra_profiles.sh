#!/bin/bash
# Run ON THE GPU BOX (through gpurun), after tools/collect_profiles.sh: kernel-trace stats of the eigensolver paths
# (exact finalize, faithful per-block loop) and of the small-side blocks.   usage: tools/collect_extra_profiles.sh <tag>
set -u
TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/finalize_trace -o f -- python $R/tools/finalize_trace.py 100 6 both > $OUT/finalize_trace.log 2>&1
for dd in 32768 131072; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/smallside_$dd -o s -- python $R/tools/smallside_probe.py $dd 2000 80 12 bf16x6 > $OUT/smallside_$dd.log 2>&1
  python $R/tools/smallside_probe.py $dd 2000 80 12 f32 > $OUT/smallside_${dd}_f32.log 2>&1
  python $R/tools/smallside_probe.py $dd 2000 80 12 bf16x3 > $OUT/smallside_${dd}_bf16x3.log 2>&1
done
grep -h -E "finalize|faithful" $OUT/finalize_trace.log | cut -c1-300
tail -n 2 $OUT/smallside_*.log
