#!/bin/bash
# round 5: bench.py with default flags from the final source (the line profiles/r05_bench.json holds)
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r05l; mkdir -p $O
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05l/bench.json"))
print("value",d["value"],"ms/step",d["ms_per_step"],"roofline",d["roofline"]["frac"])
w=d.get("wide_feature_shapes",{})
for k,v in w.items(): print(k, v["f32"]["ms_per_block"], v["bf16x6"]["ms_per_block"], v.get("cpu_baseline",{}).get("steady_blocks_timed"))
e=d.get("end_to_end_cfg3_cfg5",{})
for k,v in e.items(): print(k, json.dumps(v.get("phases")), json.dumps(v.get("vs_exact_pca_of_all_n_activations",v.get("vs_sklearn_recurrence_at_reduced_n")))[:300])
PY
