#!/bin/bash
# Round 4: window groups, second look -- tail waves at raised issue priority, mean-length pieces per group, the second
# stream at the lowest dispatch priority.  Same-box A/B KYB_MSM_GROUPS=1 / 2 + per-stage traces.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_msm2; mkdir -p $O; export TMPDIR=/tmp
for g in 1 2 1 2; do
  KYB_MSM_GROUPS=$g timeout 200 python tools/msm_bls_probe.py 1048576 15 2>/dev/null | tail -1 | tee -a $O/ab.jsonl
done
for sub in 32 64; do KYB_MSM_SUB=$sub KYB_MSM_GROUPS=2 timeout 200 python tools/msm_bls_probe.py 1048576 15 2>/dev/null | tail -1 | tee -a $O/ab_sub$sub.jsonl; done
for g in 1 2; do
  KYB_MSM_GROUPS=$g timeout 200 rocprofv3 --kernel-trace --stats -d $O -o msm_g${g}_trace -- python tools/msm_bls_probe.py 1048576 10 > $O/msm_g${g}_trace.log 2>&1
done
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
head -9 $O/msm_g2_trace.txt | cut -c1-150; head -9 $O/msm_g1_trace.txt | cut -c1-150
