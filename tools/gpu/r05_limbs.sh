#!/bin/bash
# Round 5: bucket pieces and fixed-base walks in limb form (fp_limbs.cuh, curve.cuh XyzzL): parity tests, then a same-box
# A/B against the packed form (libkyberhip_packed.so = -DKYB_XYZZ_PACKED on the two units) and a per-stage trace.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_limbs; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_full_size.py tests/test_gpu_fixed_base.py tests/test_gpu_callers.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for rep in 1 2; do
  for lib in "" kyber_amd/lib/libkyberhip_packed.so; do
    KYBER_HIP_LIB=$lib timeout 200 python tools/msm_bls_probe.py 1048576 15 all 2>/dev/null | tail -1 | sed "s|^{|{\"lib\": \"${lib:-limbs}\", |" | tee -a $O/ab.jsonl
    KYBER_HIP_LIB=$lib timeout 200 python tools/fb_probe.py bls12381 1048576 2>/dev/null | tail -1 | sed "s|^{|{\"lib\": \"${lib:-limbs}\", |" | tee -a $O/ab.jsonl
  done
done
timeout 200 rocprofv3 --kernel-trace --stats -d $O -o msm_trace -- python tools/msm_bls_probe.py 1048576 10 > $O/msm_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
head -12 $O/msm_trace.txt | cut -c1-150
