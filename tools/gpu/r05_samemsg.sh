#!/bin/bash
# Round 5: same-message verification (tbls.Recover's loop) + the round's other new tests, then its timing against the
# general entry point with the message repeated.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_samemsg; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_verify_same_msg.py tests/test_gpu_verify_same_key.py tests/test_gpu_callers.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
timeout 300 python - > $O/timing.json 2> $O/timing.err <<'P'
import hashlib, json, numpy as np, torch, bench
from kyber_amd.pairing import bls12381 as m
n = 1 << 16
k = torch.from_numpy(bench.be_scalars(b"sm/k", n)).cuda()
X, _ = m.g2_commit(k)
msg = hashlib.sha256(b"m").digest()
Hm = torch.from_numpy(np.asarray(m.batch_hash_g1([msg])[0])).cuda().repeat(n, 1)
sig, _ = m.g1_batch_mul(k, Hm)
msgs = torch.from_numpy(np.frombuffer(msg, dtype=np.uint8).copy()).cuda().repeat(n, 1)
m1 = torch.from_numpy(np.frombuffer(msg, dtype=np.uint8).copy()).cuda()
res = {"n": n}
for name, fl in (("flags0", 0), ("keys_trusted", m.F_TRUSTED(0))):
    res["general_ms_" + name] = bench.timed(lambda: m.batch_verify_g1(X, msgs, sig, flags=fl))
    res["same_msg_ms_" + name] = bench.timed(lambda: m.batch_verify_g1_same_msg(X, m1, sig, flags=fl))
ok, st = m.batch_verify_g1_same_msg(X, m1, sig)
res["all_true"] = bool(ok.all().item()) and not bool(st.any().item())
print(json.dumps(res))
P
cat $O/timing.json; tail -3 $O/timing.err
cat > /tmp/sm_probe.py <<'P'
import hashlib, json, numpy as np, torch, bench, sys
from kyber_amd.pairing import bls12381 as m
n = 1 << 16
k = torch.from_numpy(bench.be_scalars(b"sm/k", n)).cuda()
X, _ = m.g2_commit(k)
msg = hashlib.sha256(b"m").digest()
Hm = torch.from_numpy(np.asarray(m.batch_hash_g1([msg])[0])).cuda().repeat(n, 1)
sig, _ = m.g1_batch_mul(k, Hm)
msgs = torch.from_numpy(np.frombuffer(msg, dtype=np.uint8).copy()).cuda().repeat(n, 1)
m1 = torch.from_numpy(np.frombuffer(msg, dtype=np.uint8).copy()).cuda()
for _ in range(5):
    m.batch_verify_g1(X, msgs, sig); m.batch_verify_g1_same_msg(X, m1, sig)
torch.cuda.synchronize()
P
timeout 200 rocprofv3 --kernel-trace --stats -d $O -o trace -- python /tmp/sm_probe.py > $O/trace.log 2>&1
for f in $O/*.db; do python tools/rocpd_summary.py $f > ${f%_results.db}.txt 2>&1; rm -f $f; done
grep -h "operand\|tvm_kernel" $O/trace.txt | cut -c1-160
