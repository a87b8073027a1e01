#!/bin/bash
# Round 3, MSM bucket pieces in XYZZ form: GPU tests of the MSM paths, same-box A/B against the previous library
# (kyber_amd/lib/libkyberhip_msmbefore.so, built by hand from the parent commit's bls12381_msm.o), per-stage trace.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_msm_xyzz; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_full_size.py tests/test_gpu_callers.py tests/test_gpu_devices.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for i in 1 2; do
  KYBER_HIP_LIB=$PWD/kyber_amd/lib/libkyberhip_msmbefore.so timeout 300 python tools/msm_bls_probe.py 2>/dev/null | tail -1 | tee -a $O/ab_before.jsonl
  timeout 300 python tools/msm_bls_probe.py 2>/dev/null | tail -1 | tee -a $O/ab_after.jsonl
done
timeout 300 python tools/msm_probe.py 2>/dev/null | tail -1 | tee $O/msm_probe_2p20.json
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o msm_bls_trace -- python tools/msm_bls_probe.py > $O/msm_bls_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
head -20 $O/msm_bls_trace.txt
