#!/bin/bash
# Round 5: buckets per reduce chain (KYB_MSM_CHUNK) with the cooperative tail: 8 = two waves per SIMD of 41 steps, 16 = one of 57.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_chunk; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2; do for c in 4 8 16 32; do
  KYB_MSM_CHUNK=$c timeout 200 python tools/msm_bls_probe.py 1048576 15 affine 2>/dev/null | tail -1 | sed "s|^{|{\"chunk\": $c, |" | tee -a $O/ab.jsonl
done; done
for c in 16 32; do
KYB_MSM_CHUNK=$c timeout 200 rocprofv3 --kernel-trace --stats -d $O -o c$c -- python tools/msm_bls_probe.py 1048576 10 > $O/c$c.log 2>&1
done
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep -h "reduce_coop\|tree_fold\|final" $O/c*.txt | cut -c1-150
