#!/bin/bash
# (Experiment of record: the per-kind instantiation / KYB_OPERAND_WAVES switch it exercised was NOT adopted and is no
# longer in the sources -- profiles/r03_operand_kernel_experiments.json.)
# Round 3: operand kernel instantiated per set of operand kinds (default build, one wave per SIMD) against the single
# kernel that carried every path (libkyberhip_msmbefore.so holds the previous bls12381_prep.o), same box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_operand_kinds; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2; do
  for lib in msmbefore ""; do
    L=$PWD/kyber_amd/lib/libkyberhip${lib:+_$lib}.so
    KYBER_HIP_LIB=$L timeout 300 python tools/pair_probe.py bls12381 65536 2>/dev/null | tail -1 | sed "s/^{/{\"lib\": \"${lib:-default}\", /" | tee -a $O/pair.jsonl
    KYBER_HIP_LIB=$L timeout 300 python tools/verify_probe.py 65536 2>/dev/null | tail -1 | sed "s/^{/{\"lib\": \"${lib:-default}\", /" | tee -a $O/verify.jsonl
  done
done
timeout 600 python -m pytest tests/test_gpu_bls12381.py tests/test_gpu_g2_member_in_loop.py -m gpu -q -x 2>&1 | tail -2
