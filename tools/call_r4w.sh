#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python bench.py > $O/r04_bench_line.json 2> $O/r04_bench_err.log; echo "bench rc=$?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r04_bench_line.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], json.dumps(j["roofline"], indent=0)[:1500])
print(json.dumps(j.get("other_workloads"), indent=0)[:3000])
print(json.dumps(j.get("cpu_baseline"))[:600])
PY
timeout 1200 bash tools/profile_round4.sh r04 2>&1 | tail -60
