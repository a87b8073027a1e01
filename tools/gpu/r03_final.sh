#!/bin/bash
# Round-3 evidence: all GPU tests, smoke, kernel trace + separate PMC passes of the dominant kernels (Ed25519 headline,
# the pairing machine, the lane machine's G1 / G2 multiplication against the per-lane kernels it replaces, the MSM), the
# bench line.  Outputs under gpurun_out/r03_final; copy the .txt / .json summaries to profiles/r03_final_* and run
# tools/roofline_inputs.py profiles r03_final.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_final; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 300 python -c "
import sys, os; sys.path.insert(0, os.getcwd())
import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
SQ="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
prof() {  # prof <name> <command...>: trace + three PMC passes
  n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $O -o ${n}_trace -- "$@" > $O/${n}_trace.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o ${n}_fetch -- "$@" > $O/${n}_fetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o ${n}_write -- "$@" > $O/${n}_write.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $SQ -d $O -o ${n}_sq -- "$@" > $O/${n}_sq.log 2>&1
}
prof ed python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other --no-host-path
prof mul python tools/mul_probe.py bls12381 65536 3
KYB_LVM_MIN=1000000000 prof mulperlane python tools/mul_probe.py bls12381 65536 3
for s in "bls12381 65536" "bn256 262144"; do
  set -- $s
  timeout 300 python tools/pair_probe.py $1 $2 2>/dev/null | tail -1 | tee $O/probe_$1.json
  prof $1 python tools/pair_probe.py $1 $2
done
timeout 300 python tools/mul_probe.py bls12381 65536 2>/dev/null | tail -1 | tee $O/mul_probe_65536.json
timeout 300 python tools/mul_probe.py bls12381 262144 2>/dev/null | tail -1 | tee $O/mul_probe_262144.json
KYB_LVM_MIN=1000000000 timeout 300 python tools/mul_probe.py bls12381 65536 2>/dev/null | tail -1 | tee $O/mul_probe_65536_perlane.json
KYB_LVM_MIN=1000000000 timeout 300 python tools/mul_probe.py bls12381 262144 2>/dev/null | tail -1 | tee $O/mul_probe_262144_perlane.json
prof msm_bls python tools/msm_bls_probe.py
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; head -c 400 $O/bench.json
