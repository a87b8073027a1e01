#!/bin/bash
# Round 3: the endomorphism splits divide by reciprocal (divmod_z) and the MSM decode kernel stores its digits as they
# are produced.  GPU tests of everything that splits scalars (MSM, G1 / G2 Mul on both paths), same-box A/B of the
# 2^20-point MSM against the parent commit's bls12381_msm.o (libkyberhip_msmbefore.so), per-stage trace.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_msm_barrett; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_full_size.py tests/test_gpu_lane_vm.py tests/test_gpu_bls12381.py tests/test_gpu_callers.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for i in 1 2; do
  KYBER_HIP_LIB=$PWD/kyber_amd/lib/libkyberhip_msmbefore.so timeout 300 python tools/msm_bls_probe.py 2>/dev/null | tail -1 | tee -a $O/ab_before.jsonl
  timeout 300 python tools/msm_bls_probe.py 2>/dev/null | tail -1 | tee -a $O/ab_after.jsonl
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o msm_bls_trace -- python tools/msm_bls_probe.py > $O/msm_bls_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep -E "msm::|lvm_prep" $O/msm_bls_trace.txt | head -12
timeout 300 python tools/mul_probe.py bls12381 65536 2>/dev/null | tail -1 | tee $O/mul_probe_65536.json
