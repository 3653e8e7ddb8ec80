#!/bin/bash
# round 5, final call: the whole GPU suite + smoke from the final source, bench.py with default flags, kernel trace and PMC
# passes of the headline job (rocprofv3; --pmc never together with other tracing domains)
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r05f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05f/bench.json"))
print("value",d["value"],"ms/step",d["ms_per_step"],"roofline",d["roofline"]["frac"],d["roofline"].get("in_job_avg_launch_us"))
print("breakdown",d["breakdown"])
print("cpu", json.dumps(d.get("cpu_baseline"))[:200], d.get("vs_cpu_baseline"))
print("faithful", json.dumps(d.get("faithful_mode_same_job")))
print("e2e", json.dumps(d.get("end_to_end"))[:400])
w=d.get("wide_feature_shapes",{})
for k,v in w.items(): print(k, v.get("f32"), v.get("bf16x6",{}).get("ms_per_block"), json.dumps(v.get("cpu_baseline"))[:160], v.get("vs_sklearn_at_reduced_n"))
e=d.get("end_to_end_cfg3_cfg5",{})
for k,v in e.items(): print(k, json.dumps(v)[:900])
PY
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -o b -- python bench.py --no-cpu-baseline --no-wide --no-e2e > $O/bench_profiled.json 2> /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_f32_$c -o p -- python tools/gram_probe.py 131072 512 f32 > /dev/null 2>&1
done
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_f32_sq -o p -- python tools/gram_probe.py 131072 512 f32 > /dev/null 2>&1
python tools/summarize_r04.py $O > $O/summary_gram.md 2> $O/summary.err; head -40 $O/summary_gram.md
python tools/job_timeline.py $O/bench_trace > $O/job_timeline.md 2>&1; tail -14 $O/job_timeline.md
