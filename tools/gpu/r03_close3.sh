#!/bin/bash
# Last closing check of round 3: all GPU tests, smoke, the default bench line.  Every step under its own timeout.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_close5; mkdir -p $O; export TMPDIR=/tmp
timeout 200 python -m pytest tests -m gpu -q --timeout 60 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 100 python -c "
import sys, os; sys.path.insert(0, os.getcwd())
import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; head -c 200 $O/bench.json; echo
