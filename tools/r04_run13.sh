#!/bin/bash
# round 4, call 13: CholeskyQR2 range finder, lazy aux stream for one-shot Gram calls, fold test on the measure build
mkdir -p gpurun_out
{
timeout 500 python -m pytest tests/test_gpu_whole_matrix.py -q -x 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -5
timeout 200 python -m pytest tests/test_gpu_collective_shim.py -q -x 2>&1 | tail -3
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
} > gpurun_out/r04_run13.log 2>&1
tail -40 gpurun_out/r04_run13.log
