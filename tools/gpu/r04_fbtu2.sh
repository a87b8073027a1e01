#!/bin/bash
# bls12381_fb.hip on its two-wave budget (shipped) against the same unit compiled loose (libkyberhip_fbw1.so): same box
cd /root/repo; mkdir -p gpurun_out/r04_fbtu; O=gpurun_out/r04_fbtu
for lib in "" libkyberhip_fbw1.so "" libkyberhip_fbw1.so; do
  KYBER_HIP_LIB=${lib:+/root/repo/kyber_amd/lib/$lib} timeout 300 python tools/fb_probe.py bls12381 1048576 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'lib':'${lib:-shipped}',**{k:round(v,3) for k,v in d.items() if k.endswith('_ms')}}))" | tee -a $O/ab.jsonl
done
