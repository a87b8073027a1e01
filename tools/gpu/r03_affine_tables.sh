#!/bin/bash
# Round 3: affine window tables (one shared inversion) + mixed additions in the per-lane ladders: bn256 / bn254 G1 / G2,
# BLS12-381 per-lane G1 / G2.  GPU tests of the groups, then same-box A/B against the previous objects
# (libkyberhip_msmbefore.so).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_affine_tables; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_bn256.py tests/test_gpu_bn254.py tests/test_gpu_bls12381.py tests/test_gpu_full_size.py tests/test_gpu_lane_vm.py tests/test_gpu_group_conformance.py tests/test_gpu_soak.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for i in 1 2; do
  for lib in msmbefore ""; do
    L=$PWD/kyber_amd/lib/libkyberhip${lib:+_$lib}.so
    for s in "bn256 262144" "bn254 262144"; do
      KYBER_HIP_LIB=$L timeout 300 python tools/pair_probe.py $s 2>/dev/null | tail -1 | sed "s/^{/{\"lib\": \"${lib:-default}\", /" | tee -a $O/bn.jsonl | cut -c1-260
    done
    KYBER_HIP_LIB=$L KYB_LVM_MIN=1000000000 timeout 300 python tools/mul_probe.py bls12381 65536 2>/dev/null | tail -1 | sed "s/^{/{\"lib\": \"${lib:-default}\", /" | tee -a $O/bls_perlane.jsonl | cut -c1-400
  done
done
