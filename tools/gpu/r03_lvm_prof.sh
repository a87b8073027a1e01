#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_lvm_prof; mkdir -p $O; export TMPDIR=/tmp
for n in 65536 262144; do timeout 300 python tools/mul_probe.py bls12381 $n 2>/dev/null | tail -1 | tee $O/mul_$n.json; done
KYB_LVM_MIN=1000000000 timeout 300 python tools/mul_probe.py bls12381 65536 2>/dev/null | tail -1 | tee $O/mul_65536_old.json
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o mul_trace -- python tools/mul_probe.py bls12381 65536 3 > $O/mul_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
cat $O/mul_trace.txt | head -40
