#!/bin/bash
# Round 3: fixed-exponent powers (square roots of the decoders and of hash-to-curve) as one chain on unpacked limbs.
# The whole GPU suite, then same-box A/B against the previous objects (libkyberhip_msmbefore.so): UnmarshalBinary
# rates, Pair / ValidatePairing / Verify, checked MSM.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03_pow_limbs; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for i in 1 2; do
  for lib in msmbefore ""; do
    L=$PWD/kyber_amd/lib/libkyberhip${lib:+_$lib}.so
    KYBER_HIP_LIB=$L timeout 300 python tools/unmarshal_probe.py 1048576 2>/dev/null | tail -1 | sed "s/^{/{\"lib\": \"${lib:-default}\", /" | tee -a $O/unmarshal.jsonl
    KYBER_HIP_LIB=$L timeout 300 python tools/pair_probe.py bls12381 65536 2>/dev/null | tail -1 | sed "s/^{/{\"lib\": \"${lib:-default}\", /" | tee -a $O/pair.jsonl
    KYBER_HIP_LIB=$L timeout 300 python tools/verify_probe.py 65536 2>/dev/null | tail -1 | sed "s/^{/{\"lib\": \"${lib:-default}\", /" | tee -a $O/verify.jsonl
  done
done
