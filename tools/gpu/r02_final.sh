#!/bin/bash
# Round-2 evidence: all GPU tests, smoke, the bench line, kernel trace + separate PMC passes of the dominant kernels.
# Outputs under gpurun_out/r02_final; copy the .txt / .json summaries to profiles/r02_final_* and run
# tools/roofline_inputs.py profiles r02_final.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02_final; mkdir -p $O; export TMPDIR=/tmp
# "verifyonly": just the four rocprofv3 passes of the fused verification
if [ "$1" == "verifyonly" ]; then
SQ="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
V="python tools/verify_probe.py"
for c in "trace --stats" "fetch --pmc FETCH_SIZE" "write --pmc WRITE_SIZE" "sq --pmc $SQ"; do set -- $c; n=$1; shift; timeout 300 rocprofv3 --kernel-trace $* -d $O -o verify_$n -- $V > $O/verify_$n.log 2>&1; done
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
exit 0
fi
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 300 python -c "
import sys, os; sys.path.insert(0, os.getcwd())
import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
if [ "$1" != "noprof" ]; then
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other --no-host-path"
SQ="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o ed_trace -- $B > $O/ed_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o ed_fetch -- $B > $O/ed_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o ed_write -- $B > $O/ed_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQ -d $O -o ed_sq -- $B > $O/ed_sq.log 2>&1
for s in "bls12381 65536" "bn256 262144" "bn254 262144"; do
set -- $s; P="python tools/pair_probe.py $1 $2"
timeout 300 $P 2>/dev/null | tail -1 | tee $O/probe_$1.json
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o $1_trace -- $P > $O/$1_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o $1_fetch -- $P > $O/$1_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o $1_write -- $P > $O/$1_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQ -d $O -o $1_sq -- $P > $O/$1_sq.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS -d $O -o bls12381_lds -- python tools/pair_probe.py bls12381 65536 > $O/bls12381_lds.log 2>&1
V="python tools/verify_probe.py"   # the fused verification (bls12381_tvm_kernel<2>: the VERIFY program)
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o verify_trace -- $V > $O/verify_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o verify_fetch -- $V > $O/verify_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o verify_write -- $V > $O/verify_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQ -d $O -o verify_sq -- $V > $O/verify_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o msm_bls -- python tools/msm_bls_probe.py > $O/msm_bls_probe.json 2> $O/msm_bls.log
timeout 300 python tools/msm_probe.py 1048576 2>/dev/null | tail -1 > $O/msm_probe_2p20.json
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
fi
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cat $O/bench.json
