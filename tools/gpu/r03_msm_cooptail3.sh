#!/bin/bash
# Round 3: cooperative MSM tail, inlined routines, G1 and G2: MSM / caller / device / soak tests, then same-box A/B --
# one-lane tail (KYB_MSM_TAIL=lane) / cooperative tail / cooperative tail + slot-based final kernel (KYB_MSM_FINAL=slots).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_msm_cooptail4; mkdir -p $O; export TMPDIR=/tmp
timeout 150 python -m pytest tests/test_gpu_msm.py tests/test_gpu_full_size.py tests/test_gpu_callers.py tests/test_gpu_devices.py tests/test_gpu_soak.py tests/test_gpu_bn254.py -m gpu -q -x --timeout 40 > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
KYB_MSM_FINAL=slots timeout 100 python -m pytest tests/test_gpu_msm.py tests/test_gpu_full_size.py -m gpu -q -x --timeout 40 2>&1 | tail -1
for n in 1024 65536 1048576; do
  for v in "lane x" "coop x" "coop slots"; do
    set -- $v
    KYB_MSM_TAIL=$1 KYB_MSM_FINAL=$2 timeout 40 python tools/msm_bls_probe.py $n 2>/dev/null | tail -1 | sed "s/^{/{\"tail\": \"$1\", \"final\": \"$2\", /" | tee -a $O/ab.jsonl
  done
done
for v in "lane x" "coop x" "coop slots"; do
  set -- $v
  KYB_MSM_TAIL=$1 KYB_MSM_FINAL=$2 timeout 60 python tools/msm_probe.py 262144 2>/dev/null | tail -1 | sed "s/^{/{\"tail\": \"$1\", \"final\": \"$2\", /" | tee -a $O/msm_probe_2p18.jsonl
done
