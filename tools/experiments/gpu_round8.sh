#!/bin/bash
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
for s in bls12381 bn256; do timeout 600 python tools/pair_probe.py $s 65536 > gpurun_out/probe_$s.json 2> gpurun_out/probe_$s.err; cat gpurun_out/probe_$s.json; done
# the multi-process launch path with world size 1 (RCCL init, all-gather, barrier) -- what the driver runs for N > 1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err; tail -3 gpurun_out/bench_torchrun1.err; cat gpurun_out/bench_torchrun1.json
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
g.smoke()
PY
