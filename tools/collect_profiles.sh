#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats of the default bench.py run and
# separate PMC passes (never combined with sys/hip/hsa tracing) on the Gram probe.
# usage: tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>/...
set -u
TAG=${1:-r1}
ROWS=${GS_PROBE_ROWS:-131072}
export GS_PROBE_ROWS=$ROWS
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -o b -- python bench.py --no-cpu-baseline > $OUT/bench_profiled.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- python tools/gram_probe.py $ROWS > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o p -- python tools/gram_probe.py $ROWS > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_lds -o p -- python tools/gram_probe.py $ROWS > /dev/null 2>&1
python tools/summarize_profiles.py $OUT profiles/${TAG}_rocprof_summary.md > $OUT/summary.md
cat $OUT/bench.json
cat $OUT/summary.md
