#!/bin/bash
# quick A/B: lane-machine tests + mul probe (new, and per-lane with KYB_LVM_MIN huge)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_lvm_q; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_lane_vm.py -x -q 2>&1 | tail -3
for n in 65536 262144; do timeout 300 python tools/mul_probe.py bls12381 $n 2>/dev/null | tail -1 | tee $O/mul_$n.json; done
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o mul_trace -- python tools/mul_probe.py bls12381 65536 3 > $O/mul_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep "lvm_mul" $O/mul_trace.txt | cut -c1-110
