#!/bin/bash
# Round 3: the cooperative MSM tail with its routines inlined (LDS accesses as ds_* instead of flat_*): MSM tests,
# A/B against the one-lane tail, and a trial of the G2 instantiations (libkyberhip_coopfp2.so, -DKYB_COOP_FP2) in
# separate processes under short timeouts.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_msm_cooptail3; mkdir -p $O; export TMPDIR=/tmp
timeout 120 python -m pytest tests/test_gpu_msm.py tests/test_gpu_full_size.py tests/test_gpu_callers.py tests/test_gpu_devices.py tests/test_gpu_soak.py -m gpu -q -x --timeout 40 > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for n in 1024 65536 1048576; do
  for tail in lane coop; do
    KYB_MSM_TAIL=$tail timeout 40 python tools/msm_bls_probe.py $n 2>/dev/null | tail -1 | sed "s/^{/{\"tail\": \"$tail\", /" | tee -a $O/ab.jsonl
  done
done
timeout 60 rocprofv3 --kernel-trace --stats -d $O -o msm_bls_trace -- python tools/msm_bls_probe.py > $O/msm_bls_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep -E "msm::(reduce|tree|final)" $O/msm_bls_trace.txt | head -4
echo "G2 trial:"
KYBER_HIP_LIB=$PWD/kyber_amd/lib/libkyberhip_coopfp2.so timeout 30 python -m pytest "tests/test_gpu_msm.py::test_pairing_suite_msm_vs_oracle_small" -m gpu -q -x --timeout 12 2>&1 | tail -2
