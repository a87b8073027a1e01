#!/bin/bash
# register budget of the BLS12-381 G1 MSM's decode kernel (square root + subgroup test per point): 1 / 2 (shipped) / 3 / 4 waves
cd /root/repo; mkdir -p gpurun_out/r04_decwaves; O=gpurun_out/r04_decwaves
for lib in "" libkyberhip_dec1.so libkyberhip_dec3.so libkyberhip_dec4.so ""; do
  KYBER_HIP_LIB=${lib:+/root/repo/kyber_amd/lib/$lib} timeout 300 python tools/msm_bls_probe.py 1048576 5 all 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'lib':'${lib:-shipped}',**{k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items()}}))" | tee -a $O/ab.jsonl
done
