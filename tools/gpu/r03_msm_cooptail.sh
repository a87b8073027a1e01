#!/bin/bash
# Round 3: the G1 MSMs' tail (reduce / folds / final) on four cooperating lanes per point with LDS slots, against the
# one-lane kernels (KYB_MSM_TAIL=lane), same box and same library; MSM tests first (every step under its own short
# timeout: a first version of this script ran into a hanging kernel and burned twenty GPU-minutes).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_msm_cooptail2; mkdir -p $O; export TMPDIR=/tmp
timeout 150 python -m pytest tests/test_gpu_msm.py tests/test_gpu_full_size.py tests/test_gpu_callers.py tests/test_gpu_devices.py tests/test_gpu_soak.py -m gpu -q -x --timeout 40 > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for n in 1024 65536 1048576; do
  for tail in lane coop; do
    KYB_MSM_TAIL=$tail timeout 40 python tools/msm_bls_probe.py $n 2>/dev/null | tail -1 | sed "s/^{/{\"tail\": \"$tail\", /" | tee -a $O/ab.jsonl
  done
done
timeout 60 rocprofv3 --kernel-trace --stats -d $O -o msm_bls_trace -- python tools/msm_bls_probe.py > $O/msm_bls_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep -E "msm::" $O/msm_bls_trace.txt | head -8
