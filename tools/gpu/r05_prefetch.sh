#!/bin/bash
# Round 5: the accumulate kernel fetches the next point during the current addition (against -DKYB_MSM_NO_PREFETCH), same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_prefetch; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_full_size.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -2 $O/pytest.log
for rep in 1 2 3; do for lib in "" kyber_amd/lib/libkyberhip_nopf.so; do
  KYBER_HIP_LIB=$lib timeout 200 python tools/msm_bls_probe.py 1048576 15 affine 2>/dev/null | tail -1 | sed "s|^{|{\"lib\": \"${lib:-prefetch}\", |" | tee -a $O/ab.jsonl
done; done
timeout 200 python tools/msm_probe.py 1048576 2>/dev/null | tail -1 | tee $O/msm_probe.json
KYBER_HIP_LIB=kyber_amd/lib/libkyberhip_nopf.so timeout 200 python tools/msm_probe.py 1048576 2>/dev/null | tail -1 | tee $O/msm_probe_nopf.json
