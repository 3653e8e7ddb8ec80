#!/bin/bash
# round 4, final measurements from the final source: bench.json (default run), kernel traces of the bench and of the
# finalize / faithful chains, the job timeline.  (The PMC passes of tools/r04_profiles.sh are not repeated: the Gram
# kernels did not change after them.)
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r04v; mkdir -p $O
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -o b -- python bench.py --no-cpu-baseline --no-wide --no-e2e > $O/bench_profiled.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/finalize_trace -o f -- python tools/finalize_trace.py 100 6 both > $O/finalize_trace.log 2> /dev/null
python tools/job_timeline.py $O/bench_trace 1 > $O/job_timeline.md 2>&1; tail -12 $O/job_timeline.md
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04v/bench.json"))
print("value",d["value"],"ms/step",d["ms_per_step"],"roofline frac",d["roofline"]["frac"],d["roofline"].get("in_job_avg_launch_us"), d["roofline"]["steady_state_microbenchmark"])
print("breakdown",d["breakdown"])
print("vs_cpu", d.get("vs_cpu_baseline"), json.dumps(d.get("cpu_baseline"))[:300])
print(json.dumps(d.get("faithful_mode_same_job")))
print(json.dumps(d.get("timed_estimator_check")))
print(json.dumps(d.get("end_to_end"))[:700])
PY
grep "exact fin\|faithful" $O/finalize_trace.log | cut -c1-160 | tail -4
