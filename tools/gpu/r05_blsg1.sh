#!/bin/bash
# Round 5: BLS12-381 G1 Point.Mul of the per-lane band (2^15 < n <= 2^16) on lazy limbs against the packed ladder, same box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_blsg1; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bls12381.py tests/test_gpu_switches.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
for rep in 1 2; do
  for lib in "" kyber_amd/lib/libkyberhip_blspacked.so; do
    for n in 65536 32768; do
      KYBER_HIP_LIB=$lib timeout 200 python tools/mul_probe.py bls12381 $n 7 2>/dev/null | tail -1 | sed "s|^{|{\"lib\": \"${lib:-lazy}\", |" | tee -a $O/ab.jsonl
    done
  done
done
