#!/bin/bash
# Round 3: piece length (KYB_MSM_SUB) and reduce chunk (KYB_MSM_CHUNK) sweep of the 2^20-point BLS12-381 G1 MSM, same box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_msm_knobs; mkdir -p $O; export TMPDIR=/tmp
for sub in 64 96 128 256; do for ch in 8 4; do
  echo -n "{\"sub\": $sub, \"chunk\": $ch, \"r\": " >> $O/sweep.jsonl
  KYB_MSM_SUB=$sub KYB_MSM_CHUNK=$ch timeout 300 python tools/msm_bls_probe.py 2>/dev/null | tail -1 | tr -d '\n' >> $O/sweep.jsonl; echo "}" >> $O/sweep.jsonl
done; done
cat $O/sweep.jsonl
KYB_MSM_SUB=128 timeout 300 python tools/msm_probe.py 2>/dev/null | tail -1 | tee $O/msm_probe_sub128.json
KYB_MSM_SUB=64 timeout 300 python tools/msm_probe.py 2>/dev/null | tail -1 | tee $O/msm_probe_sub64.json
KYB_MSM_SUB=128 timeout 300 python tools/msm_probe.py 65536 2>/dev/null | tail -1 | tee $O/msm_probe_sub128_2p16.json
KYB_MSM_SUB=64 timeout 300 python tools/msm_probe.py 65536 2>/dev/null | tail -1 | tee $O/msm_probe_sub64_2p16.json
KYB_MSM_SUB=128 timeout 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_full_size.py -m gpu -q -x 2>&1 | tail -2
