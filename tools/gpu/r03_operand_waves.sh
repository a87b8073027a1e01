#!/bin/bash
# (Experiment of record: the per-kind instantiation / KYB_OPERAND_WAVES switch it exercised was NOT adopted and is no
# longer in the sources -- profiles/r03_operand_kernel_experiments.json.)
# Round 3: the pairing calls' operand kernel without the G2 r-torsion test / hash-to-G2 in its call graph and with a
# two-wave register budget (default build) against the same code at one wave per SIMD (libkyberhip_opw1.so:
# -DKYB_OPERAND_WAVES=1), same box; GPU tests of the pairing entry points first.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_operand_waves; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bls12381.py tests/test_gpu_g2_member_in_loop.py tests/test_gpu_full_size.py tests/test_gpu_callers.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for i in 1 2; do
  for lib in opw1 ""; do
    L=$PWD/kyber_amd/lib/libkyberhip${lib:+_$lib}.so
    KYBER_HIP_LIB=$L timeout 300 python tools/pair_probe.py bls12381 65536 2>/dev/null | tail -1 | sed "s/^{/{\"lib\": \"${lib:-default}\", /" | tee -a $O/pair.jsonl
    KYBER_HIP_LIB=$L timeout 300 python tools/verify_probe.py 65536 2>/dev/null | tail -1 | sed "s/^{/{\"lib\": \"${lib:-default}\", /" | tee -a $O/verify.jsonl
  done
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O -o pair_trace -- python tools/pair_probe.py bls12381 65536 > $O/pair_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep -E "operand|tvm_kernel" $O/pair_trace.txt | head
