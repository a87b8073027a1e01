#!/usr/bin/env python3
"""What a record of the lane machine costs, by kind: synthetic programs (one record kind repeated) run through
kyb_debug_bls12381_lvm_run on a full chip of lanes; microseconds per 1000 records and the implied cycles per record
per wave.  usage: lvm_microbench.py [waves_per_simd]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kyber_amd", "csrc"))
import numpy as np, torch
import gen_lane_vm as G
from kyber_amd import _lib
lib = C.CDLL(_lib.LIB_PATH)
fn = lib.kyb_debug_bls12381_lvm_run
fn.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.POINTER(C.c_float)]
wps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
f = G.bls12381_field()
L = G.L
def run(pair, build, reps_rec=1000):
    P = G.LProg(f, pair, 8 if pair else 4, "micro")
    c1 = P.const2(f.R % f.p, 0) if pair else P.const(f.R % f.p)
    for s in range(6):
        P.dot(s, [("linc", c1, 1)])
    with P.repeat(reps_rec):
        build(P)
    prog, sched = P.encode()
    consts = []
    for c in P.consts: consts += [d & 0xffffffff for d in f.balanced(c)] + [0] * (16 - f.N)
    dp = torch.tensor(np.array(prog, dtype=np.uint32).view(np.int32), device="cuda")
    ds = torch.tensor(np.array(sched, dtype=np.uint32).view(np.int32), device="cuda")
    dc = torch.tensor(np.array(consts, dtype=np.uint32).view(np.int32), device="cuda")
    us = C.c_float()
    nl = 256 * 4 * 64 * wps
    rc = fn(int(pair), nl, dp.data_ptr(), ds.data_ptr(), len(P.sched), dc.data_ptr(), 3, C.byref(us))
    assert rc == 0
    nrec = len(P.recs) - 6
    return us.value / (reps_rec * nrec) * 1000.0   # us per 1000 records
res = {"waves_per_simd": wps}
kinds = {
  "ctradd(no slot traffic)": (False, lambda P: P.op(G.OP_CTRADD, arg=0)),
  "fp_raw(one slot)": (False, lambda P: P.dot(0, [("lin", L(1))], raw=True)),
  "fp_raw(reg slot)": (False, lambda P: P.dot(5, [("lin", L(5))], raw=True)),
  "fp_mul(slot,slot)": (False, lambda P: P.dot(0, [("mul", L(1), L(2))])),
  "fp_mul(reg slot)": (False, lambda P: P.dot(5, [("mul", L(5), L(2))])),
  "fp_sqr": (False, lambda P: P.dot(0, [("sqr", L(1))])),
  "fp_mul(combo,combo)": (False, lambda P: P.dot(0, [("mul", L(1, 1, 2, -1), L(3, 1, 4, -1))])),
  "fp_mul+lin": (False, lambda P: P.dot(0, [("mul", L(1), L(2)), ("lin", L(3, -1, 4, -2))])),
  "fp_2mul": (False, lambda P: P.dot(0, [("mul", L(1), L(2)), ("mul", L(3), L(4))])),
  "fp_raw": (False, lambda P: P.dot(0, [("lin", L(1, 1, 2, -1))], raw=True)),
  "pair_mul2": (True, lambda P: P.dot(0, [("mul2", L(1), L(2))])),
  "pair_sqr2": (True, lambda P: P.dot(0, [("sqr2", L(1))])),
  "pair_mul2(combo)": (True, lambda P: P.dot(0, [("mul2", L(1, 1, 2, -1), L(3, 1, 4, -1))])),
  "pair_mul2+lin": (True, lambda P: P.dot(0, [("mul2", L(1), L(2)), ("lin", L(3, -4))])),
  "pair_2mul2": (True, lambda P: P.dot(0, [("mul2", L(1), L(2)), ("mul2", L(3), L(4))])),
  "pair_raw": (True, lambda P: P.dot(0, [("lin", L(1, 1, 2, -1))], raw=True)),
  "pair_mul(fp)": (True, lambda P: P.dot(0, [("mul", L(1), L(2))])),
}
only = os.environ.get('LVM_MICRO_ONLY')
for name, (pair, b) in kinds.items():
    if only and name != only: continue
    us = run(pair, b)
    res[name] = {"us_per_1000_records": round(us, 2), "cycles_per_record_per_wave": round(us * 1e-6 / 1000 * 2.4e9, 0)}
print(json.dumps({k: (v if not isinstance(v, dict) else v['cycles_per_record_per_wave']) for k, v in res.items()}))
