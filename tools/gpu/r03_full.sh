#!/bin/bash
# Round-3 checkpoint: all GPU tests, smoke, the bench line (self-verifying: outputs_match, CPU baselines).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_full; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
timeout 300 python -c "
import sys, os; sys.path.insert(0, os.getcwd())
import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; head -c 600 $O/bench.json
