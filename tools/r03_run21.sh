#!/bin/bash
out=gpurun_out/r03y; mkdir -p $out
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $out/smoke.log
timeout 60 python tools/aux_fold_check.py $out/aux.npy > $out/check.log 2>&1
GS_GRAM_NO_AUX_FOLD=1 timeout 60 python tools/aux_fold_check.py $out/noaux.npy >> $out/check.log 2>&1
python -c "
import numpy as np
a=np.load('$out/aux.npy'); b=np.load('$out/noaux.npy')
print('max abs diff', np.abs(a-b).max(), 'max abs', np.abs(a).max(), 'n', a.size)
" | tee -a $out/check.log
