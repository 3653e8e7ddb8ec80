#!/bin/bash
out=gpurun_out/r03u; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "resident or benchmarked_launch or plain_bf16 or exact_mode_matches or state_merge or gram_accumulate_matches or full_size or nccl_single" 2>&1 | tail -4 > $out/tests.log
cat $out/tests.log
for i in 1 2; do
python bench.py --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.4g'%d['value'],'ms/step',d['ms_per_step'],'launch_us',d['roofline']['avg_launch_us'],d['breakdown'])" | tee -a $out/bench.log
GS_GRAM_NO_AUX_FOLD=1 python bench.py --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NO_AUX value %.4g'%d['value'],'ms/step',d['ms_per_step'],'launch_us',d['roofline']['avg_launch_us'],d['breakdown'])" | tee -a $out/bench.log
done
