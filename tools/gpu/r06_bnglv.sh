#!/bin/bash
# round 6: the BN G1 MSM on GLV halves (9 windows instead of 17) -- parity, A/B against the library built from the adapter
# without the split (libkyberhip_noglv.so), trace
set -u
O=gpurun_out/r06_bnglv; mkdir -p $O; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_msm.py tests/test_gpu_bn256.py tests/test_gpu_bn254.py tests/test_gpu_callers.py tests/test_gpu_full_size.py tests/test_gpu_devices.py tests/test_gpu_soak.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
OLD=$PWD/kyber_amd/lib/libkyberhip_noglv.so
tag() { sed "s/^{/{\"run\": \"$1\", /"; }
for n in 1048576 1048576 262144 65536 4096; do
  KYBER_HIP_LIB=$OLD timeout 600 python tools/msm_probe.py $n 2>/dev/null | tail -1 | tag noglv >> $O/ab.jsonl
  timeout 600 python tools/msm_probe.py $n 2>/dev/null | tail -1 | tag glv >> $O/ab.jsonl
done
cat $O/ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['run'], d['n'], {k: round(v, 3) for k, v in d.items() if k.startswith('bn256')})"
timeout 600 rocprofv3 --kernel-trace --stats -d $O -o glv_trace -- python tools/msm_probe.py 1048576 > $O/glv_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep -E "Bn256Fp|Fp<kyb::Bn2" $O/glv_trace.txt | grep -v Fp2 | cut -c1-100,150-200
