#!/bin/bash
# Closing check of round 3's second session: all GPU tests, smoke, the default bench line, and fresh trace + PMC passes
# of what changed since r03_final (the MSM pipeline, the per-lane ladders).  Every step under its own timeout.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${CLOSE_TAG:-r03_close2}; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -q --timeout 60 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 120 python -c "
import sys, os; sys.path.insert(0, os.getcwd())
import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 420 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; head -c 300 $O/bench.json; echo
SQ="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
prof() {
  n=$1; shift
  timeout 90 rocprofv3 --kernel-trace --stats -d $O -o ${n}_trace -- "$@" > $O/${n}_trace.log 2>&1
  timeout 90 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o ${n}_fetch -- "$@" > $O/${n}_fetch.log 2>&1
  timeout 90 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o ${n}_write -- "$@" > $O/${n}_write.log 2>&1
  timeout 90 rocprofv3 --kernel-trace --pmc $SQ -d $O -o ${n}_sq -- "$@" > $O/${n}_sq.log 2>&1
}
prof msm_bls python tools/msm_bls_probe.py
KYB_LVM_MIN=1000000000 prof mulperlane python tools/mul_probe.py bls12381 65536 3
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
timeout 60 python tools/msm_probe.py 2>/dev/null | tail -1 | tee $O/msm_probe_2p20.json
timeout 60 python tools/pair_probe.py bn256 262144 2>/dev/null | tail -1 | tee $O/probe_bn256.json | cut -c1-200
