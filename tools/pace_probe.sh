cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gram" > gpurun_out/t.log 2>&1; grep -E "passed|failed" gpurun_out/t.log
for np in 0 1; do
  if [ $np = 1 ]; then export GS_GRAM_NO_PACE=1; fi
  echo "no_pace=$np"; timeout 100 python tools/gram_probe.py 50000 2>&1 | grep -E "gram_partial|update" | tail -3
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pace_$np -o p -- python tools/gram_probe.py 50000 > /dev/null 2>&1
  python - <<PY
import csv,glob
f=glob.glob("gpurun_out/pace_$np/*counter_collection.csv")[0]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gram_partial" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
print("FETCH_SIZE KiB avg", sum(v)/len(v), "-> MB x2:", sum(v)/len(v)*1024*2/1e6, "n", len(v))
PY
done
