#!/bin/bash
# lane machine chunk sizes (elements per prep / mul / encode round; the window tables are 2.3 / 4.6 KB per lane):
# shipped 2^17 (G1) / 2^16 (G2) against 2^19 / 2^18, same box
cd /root/repo; mkdir -p gpurun_out/r04_chunk
for lib in "" libkyberhip_g1c19.so "" libkyberhip_g1c19.so; do
  for n in 262144 1048576; do
  KYBER_HIP_LIB=${lib:+/root/repo/kyber_amd/lib/$lib} timeout 300 python tools/mul_probe.py bls12381 $n 5 | python -c "import sys,json; d=json.load(sys.stdin); print(json.dumps({'lib':'${lib:-shipped}','n':d['n'],**{k:round(v,3) for k,v in d.items() if k.endswith('_ms')}}))" | tee -a gpurun_out/r04_chunk/ab.jsonl
  done
done
