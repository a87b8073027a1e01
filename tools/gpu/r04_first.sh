#!/bin/bash
# Round 4, first look: the whole GPU suite on the new tree (scalar Horner, switch tests, oracle sample of the fixed-base
# path), the default bench line with the composite scalars, and the forced-RCCL world-1 line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_first; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
timeout 300 python -c "
import sys, os; sys.path.insert(0, os.getcwd())
import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; tail -c 1500 $O/bench.json
KYB_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_world1_forced_dist.json 2> $O/bench_world1.err; tail -2 $O/bench_world1.err; head -c 600 $O/bench_world1_forced_dist.json
