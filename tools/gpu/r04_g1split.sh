#!/bin/bash
# G1 Mul with the r-torsion test and the multiplication in different workgroups (bls12381_g1split.hip): switch tests,
# then the same-box A/B at 2^15 and 2^16
cd /root/repo; mkdir -p gpurun_out/r04_g1split; O=gpurun_out/r04_g1split
timeout 900 python -m pytest tests/test_gpu_switches.py -q -x -k "g1split or pipe or lvm" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python -m pytest tests/test_gpu_bls12381.py -q -x > $O/tests_bls.log 2>&1; tail -2 $O/tests_bls.log
for n in 32768 65536; do
  for sw in 1 0; do
    KYB_G1_SPLIT=$sw timeout 300 python tools/mul_probe.py bls12381 $n 7 | python -c "import sys,json; d=json.load(sys.stdin); print(json.dumps({'split':$sw,'n':d['n'],**{k:round(v,3) for k,v in d.items() if k.startswith('g1') and k.endswith('_ms')}}))" | tee -a $O/ab.jsonl
  done
done
