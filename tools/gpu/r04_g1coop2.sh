#!/bin/bash
# Round 4: small-batch G1 kernel with the subgroup rule on the cooperating lanes too.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_g1coop2; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bls12381.py tests/test_gpu_lane_vm.py tests/test_gpu_switches.py tests/test_gpu_soak.py tests/test_gpu_callers.py tests/test_gpu_full_size.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for c in 0 default; do
  if [ $c = default ]; then unset KYB_G1_COOP_MAX; else export KYB_G1_COOP_MAX=$c; fi
  for n in 64 4096 16384; do
    echo "{\"coop_max\": \"$c\"," $(timeout 200 python tools/mul_probe.py bls12381 $n 9 2>/dev/null | tail -1 | cut -c2-) | tee -a $O/mul_by_size.jsonl | cut -c1-330
  done
done
unset KYB_G1_COOP_MAX
timeout 300 python tools/latency_probe.py 2>/dev/null | tail -1 > $O/latency_coop.json
