#!/bin/bash
# the BLS12-381 operand kernel (bls12381_prep.hip) and the hash kernels (bls12381_h2c.hip) on a two-wave register budget
# (512 -> 256 registers): same-box A/B of the pairing calls and the fused verifications
cd /root/repo; mkdir -p gpurun_out/r04_tuwaves3; O=gpurun_out/r04_tuwaves3
for lib in "" libkyberhip_prepw2.so; do
  L=${lib:+/root/repo/kyber_amd/lib/$lib}
  for n in 65536 16384; do
    KYBER_HIP_LIB=$L timeout 300 python tools/pair_probe.py bls12381 $n 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'lib':'${lib:-shipped}','n':d['n'],**{k:round(v,3) for k,v in d.items() if k.endswith('_ms')}}))" | tee -a $O/pair.jsonl
  done
  KYBER_HIP_LIB=$L timeout 300 python tools/verify_probe.py 65536 2>/dev/null | tail -1 | cut -c1-900 | tee -a $O/verify.jsonl
done
