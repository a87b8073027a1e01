#!/bin/bash
# Round 5 experiment: the BLS12-381 G2 r-torsion test's [z]Q on lazy limbs (fourteen per coefficient; the doubling is ~75 KB
# of straight line) against the packed code, same box; and the G1 hash with its cofactor clearing on lazy limbs.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_g2lazy; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_bls12381.py tests/test_gpu_h2c.py tests/test_gpu_verify_same_msg.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for rep in 1 2; do
  for lib in "" kyber_amd/lib/libkyberhip_g2lazy.so kyber_amd/lib/libkyberhip_blspacked.so; do
    tag="{\"lib\": \"${lib:-default}\", "
    KYBER_HIP_LIB=$lib timeout 200 python tools/mul_probe.py bls12381 65536 7 2>/dev/null | tail -1 | sed "s|^{|$tag|" | tee -a $O/ab.jsonl | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print(d['lib'][-14:], {k: round(v, 2) for k, v in d.items() if k.startswith('g2') and k.endswith('_ms')})"
    KYBER_HIP_LIB=$lib timeout 200 python tools/unmarshal_probe.py 1048576 2>/dev/null | tail -1 | sed "s|^{|$tag|" | tee -a $O/ab.jsonl | cut -c1-200
    KYBER_HIP_LIB=$lib timeout 200 python tools/verify_probe.py 65536 2>/dev/null | tail -1 | sed "s|^{|$tag|" | tee -a $O/ab.jsonl | cut -c1-400
  done
done
