#!/bin/bash
# BN G2 GLS walk with the conditional mixed addition in one out-of-line call (curve.cuh jac_madd_if): tests, mul_probe
cd /root/repo; mkdir -p gpurun_out/r04_maddif; O=gpurun_out/r04_maddif
timeout 1200 python -m pytest tests/test_gpu_bn256.py tests/test_gpu_bn254.py -q -x > $O/tests.log 2>&1; tail -2 $O/tests.log
for s in bn256 bn254; do timeout 300 python tools/mul_probe.py $s 262144 7 | python -c "import sys,json; d=json.load(sys.stdin); print(json.dumps({'suite':d['suite'],**{k:round(v,3) for k,v in d.items() if k.endswith('_ms')}}))" | tee -a $O/mul.jsonl; done
