#!/bin/bash
# Round-end evidence: all GPU tests, smoke, bench, kernel trace + PMC passes of the bench and the pairing probes.
set -x
mkdir -p gpurun_out/final; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/final/pytest_gpu.log; tail -4 gpurun_out/final/pytest_gpu.log
python -c "
import sys, os; sys.path.insert(0, os.getcwd())
import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; tail -1 gpurun_out/final/smoke.log
timeout 900 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; tail -2 gpurun_out/final/bench.err; cat gpurun_out/final/bench.json
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-path"
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/final -o bench_trace -- $B > gpurun_out/final/bench_trace.log 2>&1
B2="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other --no-host-path"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/final -o ed_fetch -- $B2 > gpurun_out/final/ed_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/final -o ed_write -- $B2 > gpurun_out/final/ed_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/final -o ed_sq -- $B2 > gpurun_out/final/ed_sq.log 2>&1
for s in bls12381 bn256; do
P="python tools/pair_probe.py $s 65536"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/final -o ${s}_fetch -- $P > gpurun_out/final/${s}_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/final -o ${s}_write -- $P > gpurun_out/final/${s}_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/final -o ${s}_sq -- $P > gpurun_out/final/${s}_sq.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/final -o msm_bls -- python tools/msm_bls_probe.py > gpurun_out/final/msm_bls_probe.json 2> gpurun_out/final/msm_bls.log
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/final -o msm_ed -- python tools/msm_ed_probe.py > gpurun_out/final/msm_ed_probe.json 2> gpurun_out/final/msm_ed.log
timeout 300 python tools/msm_probe.py 1048576 > gpurun_out/final/msm_probe_2p20.json 2>/dev/null
timeout 300 python tools/msm_probe.py 65536 > gpurun_out/final/msm_probe_2p16.json 2>/dev/null
for f in gpurun_out/final/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
ls gpurun_out/final | head -60
