#!/bin/bash
# Round 4: G1Elt.Mul on four cooperating lanes for small batches (bls12381_g1coop.cuh): tests, then same-box A/B
# KYB_G1_COOP_MAX=0 (off) against the default threshold -- host-buffer latency per call and resident timings by size.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_g1coop; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bls12381.py tests/test_gpu_lane_vm.py tests/test_gpu_switches.py tests/test_gpu_soak.py tests/test_gpu_callers.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for c in 0 default 0 default; do
  if [ $c = default ]; then unset KYB_G1_COOP_MAX; else export KYB_G1_COOP_MAX=$c; fi
  for n in 64 1024 4096 16384 32768; do
    echo "{\"coop_max\": \"$c\"," $(timeout 200 python tools/mul_probe.py bls12381 $n 9 2>/dev/null | tail -1 | cut -c2-) | tee -a $O/mul_by_size.jsonl | cut -c1-400
  done
done
unset KYB_G1_COOP_MAX
timeout 300 python tools/latency_probe.py 2>/dev/null | tail -1 > $O/latency_coop.json
KYB_G1_COOP_MAX=0 timeout 300 python tools/latency_probe.py 2>/dev/null | tail -1 > $O/latency_off.json
python - <<'PY'
import json
a=json.load(open("gpurun_out/r04_g1coop/latency_coop.json")); b=json.load(open("gpurun_out/r04_g1coop/latency_off.json"))
print("sizes", a["sizes"]); print("g1 coop", a["latency_us"]["bls12381_g1_mul_validated"]); print("g1 off ", b["latency_us"]["bls12381_g1_mul_validated"])
PY
