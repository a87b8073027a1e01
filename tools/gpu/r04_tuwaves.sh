#!/bin/bash
# bn256 / bn254 scalar-multiplication translation units with EVERY kernel on a two- (three-) wave register budget, so that
# the shared out-of-line callees are compiled for it (kernel_regs: g2_mul 374 -> 256 / 168 registers): same-box A/B
cd /root/repo; mkdir -p gpurun_out/r04_tuwaves; O=gpurun_out/r04_tuwaves
for lib in "" libkyberhip_tu2.so libkyberhip_tu3.so; do
  for s in bn256 bn254; do
    KYBER_HIP_LIB=${lib:+/root/repo/kyber_amd/lib/$lib} timeout 300 python tools/mul_probe.py $s 262144 7 | python -c "import sys,json; d=json.load(sys.stdin); print(json.dumps({'lib':'${lib:-shipped}','suite':d['suite'],**{k:round(v,3) for k,v in d.items() if k.endswith('_ms')}}))" | tee -a $O/ab.jsonl
  done
  KYBER_HIP_LIB=${lib:+/root/repo/kyber_amd/lib/$lib} timeout 300 python tools/fb_probe.py bn256 262144 2>/dev/null | tail -1 | cut -c1-600 | tee -a $O/fb.jsonl
done
