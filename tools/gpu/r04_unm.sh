#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_unm; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
python - <<'PY' | tee $O/unmarshal_small.json
import hashlib, json, os, sys, subprocess
code = r'''
import hashlib, json, sys, numpy as np, torch
sys.path.insert(0, ".")
from kyber_amd.pairing import bls12381 as m
def t(fn, reps=9):
    fn(); torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts)//2]
res={}
for n in (64, 4096, 8192):
    k=np.frombuffer(hashlib.shake_256(b"u").digest(n*32),dtype=np.uint8).reshape(n,32).copy(); k[:,0]&=0x3f
    kd=torch.from_numpy(k).cuda()
    g1,_=m.g1_commit(kd); g2,_=m.g2_commit(kd)
    res[n]={"g1_unmarshal_ms":round(t(lambda: m.g1_batch_unmarshal(g1)),3),"g2_unmarshal_ms":round(t(lambda: m.g2_batch_unmarshal(g2)),3)}
print(json.dumps(res))
'''
out={}
for name,env in (("per_lane",{"KYB_G1_COOP_MAX":"0","KYB_G2_COOP_MAX":"0"}),("cooperating_lanes",{})):
    e=dict(os.environ); e.update(env)
    r=subprocess.run([sys.executable,"-c",code],env=e,capture_output=True,text=True)
    out[name]=json.loads(r.stdout.strip().split("\n")[-1]) if r.returncode==0 else r.stderr[-500:]
print(json.dumps(out))
PY
