#!/bin/bash
# GT exponentiation on the tower machine: all GPU tests, the probe, kernel trace + SQ counters of the probe, the bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_gt; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 300 python tools/gt_probe.py 65536 5 2>/dev/null | tail -1 | tee $O/gt_probe_65536.json
timeout 300 python tools/gt_probe.py 16384 5 2>/dev/null | tail -1 | tee $O/gt_probe_16384.json
SQ="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o gtmul_trace -- python tools/gt_probe.py 65536 3 > $O/gtmul_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQ -d $O -o gtmul_sq -- python tools/gt_probe.py 65536 3 > $O/gtmul_sq.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; head -c 300 $O/bench.json
