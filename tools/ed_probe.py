"""Times the Ed25519 batch kernels at n elements (default 2^20) through the engine: variable-base, fixed-base, the
unmarshal check, the MSM.  One JSON line; KYBER_HIP_LIB selects the library build (same-box A/B runs)."""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from kyber_amd.group import edwards25519 as ed

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
raw = np.frombuffer(hashlib.shake_256(b"ed-probe").digest(n * 32), dtype=np.uint8).reshape(n, 32).copy()
raw[:, 31] &= 0x0F
s = torch.from_numpy(raw).cuda()
P = ed.batch_mul_base(s)
want = hashlib.sha256(ed.batch_mul(s[:4096].contiguous(), P[:4096].contiguous())[0].cpu().numpy().tobytes()).hexdigest()


def t(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    times = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b))
    return float(np.median(times))


res = {"n": n, "digest_4096": want[:16],
       "mul_ms": t(lambda: ed.batch_mul(s, P)),
       "mul_base_ms": t(lambda: ed.batch_mul_base(s)),
       "msm_ms": t(lambda: ed.msm(s, P)),
       # KYB_F_UNIFORM (tables scanned, not indexed): what the scalar-independent access pattern costs
       "mul_uniform_ms": t(lambda: ed.batch_mul(s, P, uniform=True)),
       "mul_base_uniform_ms": t(lambda: ed.batch_mul_base(s, uniform=True))}
res["uniform_outputs_equal_default"] = bool(torch.equal(ed.batch_mul_base(s, uniform=True), ed.batch_mul_base(s)) and
                                            torch.equal(ed.batch_mul(s, P, uniform=True)[0], ed.batch_mul(s, P)[0]))
print(json.dumps(res))
