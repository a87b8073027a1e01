#!/bin/bash
# bls12381.hip with every kernel on a two-wave register budget (per-lane G1 / G2 unmarshal + mul kernels 288-505 -> 256
# registers; the coop kernels too in this experiment build): same-box A/B against the shipped library
cd /root/repo; mkdir -p gpurun_out/r04_tuwaves2; O=gpurun_out/r04_tuwaves2
for lib in "" libkyberhip_blsw2.so; do
  L=${lib:+/root/repo/kyber_amd/lib/$lib}
  for n in 1024 16384 49152 65536 262144; do
    KYBER_HIP_LIB=$L timeout 300 python tools/mul_probe.py bls12381 $n 7 | python -c "import sys,json; d=json.load(sys.stdin); print(json.dumps({'lib':'${lib:-shipped}','n':d['n'],**{k:round(v,3) for k,v in d.items() if k.endswith('_ms')}}))" | tee -a $O/mul.jsonl
  done
  KYBER_HIP_LIB=$L timeout 300 python tools/unmarshal_probe.py 65536 2>/dev/null | tail -1 | tee -a $O/unm.jsonl
  KYBER_HIP_LIB=$L timeout 300 python tools/unmarshal_probe.py 1048576 2>/dev/null | tail -1 | tee -a $O/unm.jsonl
  KYBER_HIP_LIB=$L timeout 300 python tools/fb_probe.py bls12381 1048576 2>/dev/null | tail -1 | cut -c1-700 | tee -a $O/fb.jsonl
done
