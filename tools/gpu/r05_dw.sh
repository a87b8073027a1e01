#!/bin/bash
# Round 5: register budget of the MSM decode kernel (BLS12-381 G1) now that its r-torsion test runs on lazy limbs: 2 waves (default) / 1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_dw; mkdir -p $O
for rep in 1 2; do for lib in "" kyber_amd/lib/libkyberhip_dw1.so; do
  KYBER_HIP_LIB=$lib timeout 200 python tools/msm_bls_probe.py 1048576 7 all 2>/dev/null | tail -1 | sed "s|^{|{\"decode_waves\": \"${lib:-2}\", |" | tee -a $O/ab.jsonl
done; done
