#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_member2; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -12 $O/pytest_gpu.log
for s in bn254 bn256; do timeout 300 python tools/pair_probe.py $s 262144 2>/dev/null | tail -1 | tee $O/pair_probe_$s.json; done
