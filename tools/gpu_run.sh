#!/bin/bash
# The ONE script gpurun calls on the GPU box (replaces the per-call r0X_run*.sh files of rounds 3-5).
#
#   gpurun --timeout S -- 'bash tools/gpu_run.sh <tag> <step> [<step> ...]'       -> gpurun_out/<tag>/...
#
# steps (run in the order given; each writes under gpurun_out/<tag>/):
#   tests[:<pytest -k expr or file list>]   pytest -m gpu (whole suite, or a selection)        -> tests.log
#   smoke                                   __graft_entry__.smoke()                            -> smoke.log
#   bench[:<extra flags>]                   python bench.py <flags>                            -> bench.json / bench.err
#   benchprof[:<extra flags>]               rocprofv3 --kernel-trace --stats of bench.py       -> benchprof_kernel_stats.csv
#   e2e:<job>[:n[:batch]]                   tools/e2e_job.py <job> (cfg2 cfg2f cfg3 cfg4 cfg4f cfg5) -> e2e_<job>.json
#   e2ewall:<job>[:n[:batch]]               the same without the phase timers (no synchronisation between phases) -> e2ewall_<job>.json
#   e2eprof:<job>[:n[:batch]]               the same under rocprofv3 --kernel-trace --stats    -> e2eprof_<job>_kernel_stats.csv
#   py:<script>[:args...]                   python tools/<script> args (':'-separated)        -> <script>.log
#   pyprof:<script>[:args...]               the same under rocprofv3 --kernel-trace --stats    -> <script>_kernel_stats.csv
#   apiprof:<script>[:args...]              rocprofv3 --hip-trace --kernel-trace --stats       -> <script>_hip_api_stats.csv
#   pmc:<counters '+'-joined>:<script>[:args...]   one rocprofv3 --pmc pass (never combined with other tracing domains)
#   timeline                                rocprofv3 --kernel-trace of `bench.py --no-extras` -> job_timeline.md (tools/job_timeline.py)
#   grampmc[:rows]                          FETCH_SIZE / WRITE_SIZE / SQ / LDS --pmc passes on tools/gram_probe.py (separate passes)
#                                           -> rocprof_summary.md + gram_pmc_latest.json (tools/summarize_profiles.py)
#   env:<NAME>:<VALUE>                      export NAME=VALUE for the following steps
#   measure                                 export GANSPACE_HIP_LIB=lib_measure for the following steps
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p "$O"
stats() {   # <trace dir> <out csv>: keep the kernel-stats table of a rocprofv3 --stats run, drop the raw trace (gpurun merges
            # at most 64 MiB back: a kernel trace of the bench run alone is larger)
  f=$(find "$1" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$2" && head -25 "$2"
  rm -rf "$1"
}
for step in "$@"; do
  IFS=':' read -r kind a1 rest <<< "$step"
  echo "=== $step"
  case $kind in
    tests)
      if [ -z "${a1:-}" ]; then timeout 3000 python -m pytest tests -m gpu -x -q > "$O/tests.log" 2>&1
      elif [[ "$a1" == tests/* ]]; then timeout 3000 python -m pytest ${a1//,/ } -m gpu -x -q > "$O/tests.log" 2>&1
      else timeout 3000 python -m pytest tests -m gpu -x -q -k "$a1" > "$O/tests.log" 2>&1; fi
      tail -15 "$O/tests.log" ;;
    smoke) timeout 600 python __graft_entry__.py --smoke > "$O/smoke.log" 2>&1; tail -4 "$O/smoke.log" ;;
    bench)
      ( time timeout 1500 python bench.py ${a1:-} ${rest//:/ } > "$O/bench.json" 2> "$O/bench.err" ) 2>&1 | grep real
      tail -c 1500 "$O/bench.json" ;;
    benchprof)
      timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/benchprof" -o b -- python bench.py ${a1:-} ${rest//:/ } > "$O/benchprof.json" 2> "$O/benchprof.err"
      stats "$O/benchprof" "$O/benchprof_kernel_stats.csv" ;;
    e2e) timeout 1500 python tools/e2e_job.py $a1 ${rest//:/ } > "$O/e2e_$a1.json" 2> "$O/e2e_$a1.err"; cat "$O/e2e_$a1.json" ;;
    e2ewall) GS_E2E_PROFILE=0 timeout 1500 python tools/e2e_job.py $a1 ${rest//:/ } > "$O/e2ewall_$a1.json" 2> "$O/e2ewall_$a1.err"; cat "$O/e2ewall_$a1.json" ;;
    e2eprof)
      timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/e2eprof_$a1" -o e -- python tools/e2e_job.py $a1 ${rest//:/ } > "$O/e2eprof_$a1.json" 2> "$O/e2eprof_$a1.err"
      cat "$O/e2eprof_$a1.json"; stats "$O/e2eprof_$a1" "$O/e2eprof_${a1}_kernel_stats.csv" ;;
    py) timeout 1500 python tools/$a1 ${rest//:/ } > "$O/${a1%.py}.log" 2>&1; tail -40 "$O/${a1%.py}.log" ;;
    pyprof)
      timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_${a1%.py}" -o p -- python tools/$a1 ${rest//:/ } > "$O/${a1%.py}.log" 2>&1
      tail -20 "$O/${a1%.py}.log"; stats "$O/prof_${a1%.py}" "$O/${a1%.py}_kernel_stats.csv" ;;
    apiprof)   # HIP API time per call name (where host-side milliseconds of a short job go): <script>_hip_api_stats.csv
      timeout 1500 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d "$O/api_${a1%.py}" -o p -- python tools/$a1 ${rest//:/ } > "$O/${a1%.py}.log" 2>&1
      tail -8 "$O/${a1%.py}.log"
      f=$(find "$O/api_${a1%.py}" -name '*hip_api_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$O/${a1%.py}_hip_api_stats.csv" && head -16 "$f"
      # one timeline: every kernel and the HIP calls that took more than 0.1 ms, in time order (ms since the first call)
      f=$(find "$O/api_${a1%.py}" -name '*hip_api_trace.csv' | head -1); g=$(find "$O/api_${a1%.py}" -name '*kernel_trace.csv' | head -1)
      [ -n "$f" ] && [ -n "$g" ] && python - "$f" "$g" > "$O/${a1%.py}_hip_api_slow_calls.txt" <<'PYEOF'
import csv, sys
api = list(csv.DictReader(open(sys.argv[1]))); ker = list(csv.DictReader(open(sys.argv[2])))
t0 = min(int(r["Start_Timestamp"]) for r in api)
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "API    " + r["Function"]) for r in api
      if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 100_000]
ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "kernel " + r["Kernel_Name"][:60]) for r in ker]
for s, d, n in sorted(ev): print(f"{(s - t0) / 1e6:10.3f} ms  {d / 1e3:9.1f} us  {n}")
PYEOF
      tail -60 "$O/${a1%.py}_hip_api_slow_calls.txt"
      stats "$O/api_${a1%.py}" "$O/${a1%.py}_kernel_stats.csv" > /dev/null ;;
    pmc)
      IFS=':' read -r script args <<< "$rest"
      timeout 1500 rocprofv3 --pmc ${a1//+/ } --kernel-trace --output-format csv -d "$O/pmc_${a1%%+*}" -o p -- python tools/$script ${args//:/ } > /dev/null 2>&1
      python tools/pmc_table.py "$O/pmc_${a1%%+*}" | tee "$O/pmc_${a1%%+*}.txt"; rm -rf "$O/pmc_${a1%%+*}" ;;
    timeline)
      timeout 1500 rocprofv3 --kernel-trace --output-format csv -d "$O/timeline" -o t -- python bench.py --no-extras > "$O/timeline_bench.json" 2> /dev/null
      python tools/job_timeline.py "$O/timeline" > "$O/job_timeline.md"; cat "$O/job_timeline.md"; rm -rf "$O/timeline" ;;
    grampmc)
      ROWS=${a1:-131072}; export GS_PROBE_ROWS=$ROWS
      [ -f "$O/benchprof_kernel_stats.csv" ] && mkdir -p "$O/bench_trace" && cp "$O/benchprof_kernel_stats.csv" "$O/bench_trace/b_kernel_stats.csv"
      for c in FETCH_SIZE WRITE_SIZE; do
        timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_$c" -o p -- python tools/gram_probe.py $ROWS > /dev/null 2>&1
      done
      timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/pmc_sq" -o p -- python tools/gram_probe.py $ROWS > /dev/null 2>&1
      timeout 900 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d "$O/pmc_lds" -o p -- python tools/gram_probe.py $ROWS > /dev/null 2>&1
      cp profiles/gram_pmc_latest.json "$O/gram_pmc_before.json" 2>/dev/null
      python tools/summarize_profiles.py "$O" "profiles/${TAG}_rocprof_summary.md" > "$O/rocprof_summary.md"
      cp profiles/gram_pmc_latest.json "$O/gram_pmc_latest.json"; cat "$O/rocprof_summary.md"
      rm -rf "$O"/pmc_FETCH_SIZE "$O"/pmc_WRITE_SIZE "$O"/pmc_sq "$O"/pmc_lds "$O"/bench_trace ;;
    env) export "$a1=${rest}" ;;        # env:NAME:VALUE for the following steps
    measure) export GANSPACE_HIP_LIB="$R/ganspace_amd/lib_measure/libganspace_hip.so" ;;
    *) echo "unknown step $step" ;;
  esac
done
