#!/bin/bash
# Round 5: BN G1 / G2 Point.Mul on lazy 30-bit limbs (jac_lazy.cuh) against the packed ladders (-DKYB_BN_PACKED_LADDER),
# same box: parity tests first, then alternating timings at 2^18 and a trace of each.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_bnlazy; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bn256.py tests/test_gpu_bn254.py tests/test_gpu_full_size.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for rep in 1 2; do
  for lib in "" kyber_amd/lib/libkyberhip_bnpacked.so; do
    for s in bn256 bn254; do
      KYBER_HIP_LIB=$lib timeout 200 python tools/mul_probe.py $s 262144 7 2>/dev/null | tail -1 | sed "s|^{|{\"lib\": \"${lib:-lazy}\", |" | tee -a $O/ab.jsonl
    done
  done
done
timeout 200 rocprofv3 --kernel-trace --stats -d $O -o lazy -- python tools/mul_probe.py bn256 262144 5 > $O/lazy.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep -h "mul_kernel" $O/lazy.txt | cut -c1-160
