#!/bin/bash
# Full measurement visit: parity tests, bench, kernel trace, PMC passes (HBM bytes, VALU activity).
set -x
mkdir -p gpurun_out/prof; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -3 gpurun_out/bench.err; cat gpurun_out/bench.json
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other"
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ed_trace -- $B > gpurun_out/prof/ed_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof -o ed_fetch -- $B > gpurun_out/prof/ed_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof -o ed_write -- $B > gpurun_out/prof/ed_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/prof -o ed_sq -- $B > gpurun_out/prof/ed_sq.log 2>&1
for s in bls12381 bn256; do
P="python tools/pair_probe.py $s 65536"
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ${s}_trace -- $P > gpurun_out/prof/${s}_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof -o ${s}_fetch -- $P > gpurun_out/prof/${s}_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof -o ${s}_write -- $P > gpurun_out/prof/${s}_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/prof -o ${s}_sq -- $P > gpurun_out/prof/${s}_sq.log 2>&1
done
ls -la gpurun_out/prof | head -40
