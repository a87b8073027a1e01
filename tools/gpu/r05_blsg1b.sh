#!/bin/bash
# Round 5: BLS12-381 G1 ladder AND r-torsion test on lazy limbs against the packed code (-DKYB_BLS_PACKED_LADDER), same box:
# Point.Mul at 2^15 / 2^16, UnmarshalBinary at 2^20, Pair / verify at 2^16, the checked MSM at 2^20.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_blsg1b; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_bls12381.py tests/test_gpu_switches.py tests/test_gpu_msm.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for rep in 1 2; do
  for lib in "" kyber_amd/lib/libkyberhip_blspacked.so; do
    tag="{\"lib\": \"${lib:-lazy}\", "
    KYBER_HIP_LIB=$lib timeout 200 python tools/mul_probe.py bls12381 65536 7 2>/dev/null | tail -1 | sed "s|^{|$tag|" | tee -a $O/ab.jsonl | cut -c1-300
    KYBER_HIP_LIB=$lib timeout 200 python tools/unmarshal_probe.py 1048576 2>/dev/null | tail -1 | sed "s|^{|$tag|" | tee -a $O/ab.jsonl | cut -c1-400
    KYBER_HIP_LIB=$lib timeout 200 python tools/pair_probe.py bls12381 65536 2>/dev/null | tail -1 | sed "s|^{|$tag|" | tee -a $O/ab.jsonl | cut -c1-500
    KYBER_HIP_LIB=$lib timeout 200 python tools/msm_bls_probe.py 1048576 7 all 2>/dev/null | tail -1 | sed "s|^{|$tag|" | tee -a $O/ab.jsonl
  done
done
