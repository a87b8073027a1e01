#!/bin/bash
set -x
mkdir -p gpurun_out/prof3; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
for s in bls12381 bn256; do timeout 600 python tools/pair_probe.py $s 65536 > gpurun_out/probe_$s.json 2> gpurun_out/probe_$s.err; cat gpurun_out/probe_$s.json; done
KYB_BENCH_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err; tail -3 gpurun_out/bench_torchrun1.err; cut -c1-300 gpurun_out/bench_torchrun1.json; grep -o '"bls12381_g1_msm_2p20": {[^}]*}' gpurun_out/bench_torchrun1.json
timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.err; cat gpurun_out/bench.json
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof3 -o bench_trace -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof3/bench_trace.log 2>&1
