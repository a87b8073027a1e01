#!/bin/bash
# Closing check of the tree: all GPU tests, smoke, the default bench line (with the CPU baselines), clocks of the box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_close; mkdir -p $O; export TMPDIR=/tmp
rocm-smi --showclocks --showpower > $O/smi_before.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 300 python -c "
import sys, os; sys.path.insert(0, os.getcwd())
import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; head -c 300 $O/bench.json
rocm-smi --showclocks --showpower > $O/smi_after.txt 2>&1
