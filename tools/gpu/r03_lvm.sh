#!/bin/bash
# lane machine bring-up: interpreter vs simulator trace, end-to-end mul tests, then the probes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_lvm; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_lane_vm.py -x -q > $O/pytest_lvm.log 2>&1; echo "rc=$?" >> $O/pytest_lvm.log; tail -25 $O/pytest_lvm.log
for n in 65536 262144; do
  timeout 300 python tools/pair_probe.py bls12381 $n 2>/dev/null | tail -1 | tee $O/probe_bls12381_$n.json
done
KYB_LVM_MIN=1000000000 timeout 300 python tools/pair_probe.py bls12381 65536 2>/dev/null | tail -1 | tee $O/probe_bls12381_65536_old.json
