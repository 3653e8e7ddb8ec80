#!/bin/bash
# Run ON THE GPU BOX: FETCH_SIZE / WRITE_SIZE / L2 hit counters of the wide split-bf16 Gram launch (131 072 x 512 rows)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/widepmc -o p_$(echo $c | cut -d' ' -f1) -- python $R/tools/gram_probe.py ${1:-131072} 512 bf16x3 > /dev/null 2>&1
done
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/widepmc/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "wide" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(k, sum(v)/len(v), "n", len(v))
PY
