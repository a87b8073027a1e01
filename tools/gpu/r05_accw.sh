#!/bin/bash
# Round 5: register budget of the limb-form accumulate kernel (2 / 3 / 4 waves per SIMD), same-box A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_accw; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2; do
  for w in 2 3 4; do
    KYBER_HIP_LIB=kyber_amd/lib/libkyberhip_accw$w.so timeout 200 python tools/msm_bls_probe.py 1048576 15 affine 2>/dev/null | tail -1 | sed "s|^{|{\"acc_waves\": $w, |" | tee -a $O/ab.jsonl
  done
done
for w in 2 3 4; do
KYBER_HIP_LIB=kyber_amd/lib/libkyberhip_accw$w.so timeout 200 rocprofv3 --kernel-trace --stats -d $O -o w$w -- python tools/msm_bls_probe.py 1048576 10 > $O/w$w.log 2>&1
done
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep -h accumulate $O/w*.txt | cut -c1-150
