#!/bin/bash
# the whole GPU suite on the current tree (log kept under gpurun_out/r04_suite)
cd /root/repo; mkdir -p gpurun_out/r04_suite; O=gpurun_out/r04_suite
timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 > $O/tests.log 2>&1; tail -25 $O/tests.log
for n in 24576 32768; do
  for sw in 1 0; do
    KYB_G1_SPLIT=$sw timeout 300 python tools/mul_probe.py bls12381 $n 7 | python -c "import sys,json; d=json.load(sys.stdin); print(json.dumps({'split':$sw,'n':d['n'],**{k:round(v,3) for k,v in d.items() if k.startswith('g1') and k.endswith('_ms')}}))" | tee -a $O/split_ab.jsonl
  done
done
