#!/bin/bash
# Round 4: bn256 ValidatePairing as product form + zero-Miller-value fallback: tests, then A/B against KYB_BN_CHECK=two.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_bncheck; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_bn256.py tests/test_gpu_bn254.py tests/test_gpu_full_size.py tests/test_gpu_switches.py tests/test_gpu_callers.py tests/test_gpu_soak.py tests/test_gpu_bls12381.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for v in two default two default; do
  if [ $v = default ]; then unset KYB_BN_CHECK; else export KYB_BN_CHECK=$v; fi
  echo "{\"bn_check\": \"$v\"," $(timeout 300 python tools/pair_probe.py bn256 262144 2>/dev/null | tail -1 | cut -c2-) | tee -a $O/pair_probe.jsonl | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['bn_check'], {k:round(v,2) for k,v in d.items() if k.endswith('_ms')})"
done
unset KYB_BN_CHECK
timeout 300 python tools/pair_probe.py bls12381 65536 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('bls', {k:round(v,2) for k,v in d.items() if k.endswith('_ms')})"
