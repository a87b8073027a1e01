#!/bin/bash
# bn256 G2 Mul behind the membership gate (GLS for members, plain ladder otherwise); bn254's G2 test on the NAF / mixed
# multiplier: tests, then mul_probe on both suites (compare with profiles/r04_final_bench.json: 26.7 / 25.8 ms)
cd /root/repo; mkdir -p gpurun_out/r04_bng2; O=gpurun_out/r04_bng2
timeout 1200 python -m pytest tests/test_gpu_bn256.py tests/test_gpu_bn254.py tests/test_gpu_soak.py -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
for s in bn256 bn254; do timeout 300 python tools/mul_probe.py $s 262144 7 | tee -a $O/mul.jsonl; done
timeout 300 python tools/unmarshal_probe.py bn254 262144 2>&1 | tail -2 | tee -a $O/unm.jsonl
timeout 300 python tools/pair_probe.py bn254 262144 2>&1 | tail -1 | tee -a $O/pair.jsonl
