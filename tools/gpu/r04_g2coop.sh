#!/bin/bash
# Round 4: small-batch G2 kernel on cooperating lanes: tests, A/B by size.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_g2coop; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bls12381.py tests/test_gpu_lane_vm.py tests/test_gpu_switches.py tests/test_gpu_soak.py tests/test_gpu_callers.py tests/test_gpu_group_conformance.py tests/test_gpu_devices.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for c in 0 1000000; do
  export KYB_G2_COOP_MAX=$c
  for n in 64 1024 4096 8192 16384; do
    echo "{\"g2_coop_max\": \"$c\"," $(timeout 200 python tools/mul_probe.py bls12381 $n 9 2>/dev/null | tail -1 | cut -c2-) | tee -a $O/mul_by_size.jsonl | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['g2_coop_max'], d['n'], {k:round(v,2) for k,v in d.items() if k.startswith('g2') and k.endswith('ms')})"
  done
done
