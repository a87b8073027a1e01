#!/bin/bash
# One GPU-box visit: parity tests, VALU micro-benchmark, bench, rocprof kernel stats.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "Name:|Compute Unit|Max Clock" | head -20 > gpurun_out/rocminfo.txt 2>&1
nproc > gpurun_out/nproc.txt; lscpu | head -20 >> gpurun_out/nproc.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o /tmp/valu_peak && timeout 120 /tmp/valu_peak > gpurun_out/valu_peak.json 2>&1
cat gpurun_out/valu_peak.json
timeout 600 python bench.py --steps 5 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -3 gpurun_out/bench.err; cat gpurun_out/bench.json
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r01 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
ls -R gpurun_out/prof | head -30
