#!/bin/bash
# round 6: the BLS12-381 G2 MSM on balanced GLS quarters (4 windows instead of 18) -- parity, A/B against the library built
# from the adapter without the split (libkyberhip_nogls.so), trace
set -u
O=gpurun_out/r06_g2gls; mkdir -p $O; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_msm.py tests/test_gpu_bls12381.py tests/test_gpu_callers.py tests/test_gpu_full_size.py tests/test_gpu_devices.py tests/test_gpu_soak.py tests/test_gpu_switches.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
OLD=$PWD/kyber_amd/lib/libkyberhip_nogls.so
cat > /tmp/g2probe.py <<'PY'
import hashlib, json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from kyber_amd.pairing import bls12381 as m
n = int(sys.argv[1])
def sc(label, n):
    a = np.frombuffer(hashlib.shake_256(label).digest(n * 32), dtype=np.uint8).reshape(n, 32).copy(); a[:, 0] &= 0x3F
    return a
k = torch.from_numpy(sc(b"k", n)).cuda(); h = torch.from_numpy(sc(b"h", n)).cuda()
g2b = torch.from_numpy(np.frombuffer(m.G2_BASE, dtype=np.uint8).copy()).cuda()
P, _ = m._mul(2, h, g2b, True); Pu, _ = m._mul(2, h, g2b, True, m.F_UNCOMPRESSED_OUT)
def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
res = {"n": n, "g2_msm_checked_ms": t(lambda: m.g2_msm(k, P)), "g2_msm_trusted_ms": t(lambda: m.g2_msm(k, P, m.F_TRUSTED(0))),
       "g2_msm_trusted_unc_ms": t(lambda: m.g2_msm(k, Pu, m.F_TRUSTED(0) | m.F_UNCOMPRESSED)),
       "g2_msm_128bit_trusted_unc_ms": t(lambda: m.g2_msm(k, Pu, m.F_TRUSTED(0) | m.F_UNCOMPRESSED | m.F_SCALAR_BITS(128)))}
a = m.g2_msm(k, P)[0]; b = m.g2_msm(k, Pu, m.F_TRUSTED(0) | m.F_UNCOMPRESSED)[0]
res["conventions_agree"] = bool((a == b).all().item())
print(json.dumps(res))
PY
tag() { sed "s/^{/{\"run\": \"$1\", /"; }
for n in 262144 262144 65536 4096 1048576; do
  KYBER_HIP_LIB=$OLD timeout 600 python /tmp/g2probe.py $n 2>/dev/null | tail -1 | tag nogls >> $O/ab.jsonl
  timeout 600 python /tmp/g2probe.py $n 2>/dev/null | tail -1 | tag gls >> $O/ab.jsonl
done
cat $O/ab.jsonl
timeout 600 rocprofv3 --kernel-trace --stats -d $O -o gls_trace -- python /tmp/g2probe.py 262144 > $O/gls_trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep -E "BlsG2Msm" $O/gls_trace.txt | cut -c1-90,130-200
