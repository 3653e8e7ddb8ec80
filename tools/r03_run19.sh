#!/bin/bash
out=gpurun_out/r03w; mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/bench.err
timeout 330 python -m pytest tests/test_gpu_decomposition.py tests/test_gpu_distributed.py tests/test_gpu_whole_matrix.py -q -m gpu -x -k "cfg2_full or whole_matrix or pca_estimator or integration or cfg1 or sharded_exact or bench_two_ranks" > $out/tests.log 2>&1
grep -E "passed|failed|error" $out/tests.log | tail -3
grep -E "^FAILED|^ERROR" $out/tests.log | head -10
tail -2 $out/bench.err | cut -c1-200
