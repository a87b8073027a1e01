#!/bin/bash
# call-flags check: full GPU parity suite + bench
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/flags_tests.log 2>&1
tail -15 gpurun_out/flags_tests.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/flags_bench.json 2> gpurun_out/flags_bench.err
tail -3 gpurun_out/flags_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/flags_bench.json"))
print(d["value"], json.dumps(d["other_workloads"], indent=1))
PY
