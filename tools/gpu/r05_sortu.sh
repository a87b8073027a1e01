#!/bin/bash
# Round 5: digits in flight per thread in the MSM's two sort kernels (KYB_MSM_SORT_U = 1 is the old one-per-trip loop), same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_sortu; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_full_size.py tests/test_gpu_ed25519.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -2 $O/pytest.log
for rep in 1 2; do for u in 8 1 4 16; do
  lib=kyber_amd/lib/libkyberhip_sortu$u.so; [ $u = 8 ] && lib=""
  KYBER_HIP_LIB=$lib timeout 200 python tools/msm_bls_probe.py 1048576 15 affine 2>/dev/null | tail -1 | sed "s|^{|{\"sort_u\": $u, |" | tee -a $O/ab.jsonl
done; done
timeout 200 rocprofv3 --kernel-trace --stats -d $O -o u8 -- python tools/msm_bls_probe.py 1048576 10 > $O/u8.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep -h "hist_lds\|scatter_lds" $O/u8.txt | cut -c1-150
timeout 200 python tools/msm_probe.py 1048576 2>/dev/null | tail -1
