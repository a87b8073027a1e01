#!/bin/bash
# Round 4: the MSM's window groups on two streams (msm.cuh run()): tests, then same-box A/B KYB_MSM_GROUPS=1 / 2 and a
# per-stage trace of each.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_msm; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_full_size.py tests/test_gpu_switches.py tests/test_gpu_callers.py tests/test_gpu_devices.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for g in 1 2 1 2; do
  for n in 1048576 65536; do
    KYB_MSM_GROUPS=$g timeout 200 python tools/msm_bls_probe.py $n 15 all 2>/dev/null | tail -1 | tee -a $O/ab.jsonl
  done
done
for g in 1 2; do
  KYB_MSM_GROUPS=$g timeout 200 python tools/msm_probe.py 1048576 2>/dev/null | tail -1 | tee -a $O/ab_all.jsonl
  KYB_MSM_GROUPS=$g timeout 200 rocprofv3 --kernel-trace --stats -d $O -o msm_g${g}_trace -- python tools/msm_bls_probe.py 1048576 10 > $O/msm_g${g}_trace.log 2>&1
done
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
head -24 $O/msm_g2_trace.txt | cut -c1-150
