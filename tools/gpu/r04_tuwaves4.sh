#!/bin/bash
# shipped tree: BN units on the two-wave budget, BLS12-381 UnmarshalBinary of large batches through bls12381_unm2.hip
# (KYB_UNM_W2=0: the loose kernels) -- tests of the suites touched, then the A/B
cd /root/repo; mkdir -p gpurun_out/r04_tuwaves4; O=gpurun_out/r04_tuwaves4
timeout 1500 python -m pytest tests/test_gpu_bn256.py tests/test_gpu_bn254.py tests/test_gpu_bls12381.py tests/test_gpu_full_size.py tests/test_gpu_lane_vm.py tests/test_gpu_fixed_base.py tests/test_gpu_msm.py -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
for sw in 1 0; do
  for n in 131072 262144 1048576; do
    KYB_UNM_W2=$sw timeout 300 python tools/unmarshal_probe.py $n 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'unm_w2':$sw,'n':d['n'],'g1_per_s':d['bls12381_g1'],'g2_per_s':d['bls12381_g2']}))" | tee -a $O/unm.jsonl
  done
  KYB_UNM_W2=$sw timeout 300 python tools/mul_probe.py bls12381 262144 7 | python -c "import sys,json; d=json.load(sys.stdin); print(json.dumps({'unm_w2':$sw,'n':d['n'],**{k:round(v,3) for k,v in d.items() if k.endswith('_ms')}}))" | tee -a $O/mul.jsonl
done
for s in bn256 bn254; do timeout 300 python tools/pair_probe.py $s 262144 2>/dev/null | tail -1 | tee -a $O/bnpair.jsonl; done
