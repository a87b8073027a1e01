#!/bin/bash
# bn256 unit on two waves (shipped) against loose (libkyberhip_bnw1.so) at SMALL batches: one wave per SIMD and below
cd /root/repo; mkdir -p gpurun_out/r04_bnsmall; O=gpurun_out/r04_bnsmall
for lib in "" libkyberhip_bnw1.so; do for n in 4096 32768 65536 131072; do
  KYBER_HIP_LIB=${lib:+/root/repo/kyber_amd/lib/$lib} timeout 300 python tools/mul_probe.py bn256 $n 7 | python -c "import sys,json; d=json.load(sys.stdin); print(json.dumps({'lib':'${lib:-shipped}','n':d['n'],**{k:round(v,3) for k,v in d.items() if k.endswith('_ms')}}))" | tee -a $O/ab.jsonl
done; done
