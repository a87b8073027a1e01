#!/bin/bash
# G2 membership decided at the end of the Miller loop: all GPU tests, pair probe, kernel trace + SQ counters, bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_member; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 300 python tools/pair_probe.py bls12381 65536 2>/dev/null | tail -1 | tee $O/pair_probe.json
SQ="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o bls12381_trace -- python tools/pair_probe.py bls12381 65536 > $O/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQ -d $O -o bls12381_sq -- python tools/pair_probe.py bls12381 65536 > $O/sq.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; head -c 300 $O/bench.json
