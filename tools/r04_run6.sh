#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r04f; mkdir -p $O
M=$R/ganspace_amd/lib_measure/libganspace_hip.so
GANSPACE_HIP_LIB=$M GS_TOPK_DEBUG=1 timeout 300 python tools/finalize_trace.py 10 2 exact 2>&1 | grep "chol_inv" | head -4
timeout 300 python tools/finalize_trace.py 100 3 both > $O/finalize.log 2>&1; grep "exact fin\|faithful" $O/finalize.log | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_topk.py tests/test_gpu_collective_shim.py -x -q > $O/t1.log 2>&1; echo "rc=$?"; grep -E "passed|failed|error|Error" $O/t1.log | tail -5
( time timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real; echo "bench rc=$?"; tail -5 $O/bench.err | cut -c1-300
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04f/bench.json"))
print("value",d["value"],"ms/step",d["ms_per_step"],"roofline frac",d["roofline"]["frac"],d["roofline"].get("in_job_avg_launch_us"),d["roofline"]["steady_state_microbenchmark"])
print("breakdown",d["breakdown"])
for k in ("cpu_baseline","faithful_mode_same_job","end_to_end","end_to_end_cfg3_cfg5"):
    print(k, json.dumps(d.get(k))[:1500])
for k,v in d.get("wide_feature_shapes",{}).items():
    print(k, v.get("ms_per_block"), json.dumps(v.get("cpu_baseline"))[:400], json.dumps(v.get("vs_sklearn_at_reduced_n")))
PY
