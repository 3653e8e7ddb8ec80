#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
bash tools/r04_profiles.sh > gpurun_out/r04p_console.log 2>&1
O=gpurun_out/r04p
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04p/bench.json"))
print("value",d["value"],"ms/step",d["ms_per_step"],"roofline frac",d["roofline"]["frac"],d["roofline"].get("in_job_avg_launch_us"), d["roofline"]["steady_state_microbenchmark"])
print("breakdown",d["breakdown"])
print("vs_cpu", d.get("vs_cpu_baseline"), json.dumps(d.get("cpu_baseline"))[:300])
print(json.dumps(d.get("faithful_mode_same_job")))
print(json.dumps(d.get("timed_estimator_check")))
print(json.dumps(d.get("cos_sim_vs_reference"))[:600])
print(json.dumps(d.get("end_to_end"))[:500])
PY
