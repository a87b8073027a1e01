"""GT exponentiation on the tower machine (program GTMUL; kyb_*_gt_mul): GTElt.Mul (kilic/gt.go:79-84) with the
order-r membership decided by the program's three comparisons, pointGT.Mul -> gfP12.Exp (pairing/bn256/point.go:613,
gfp12.go:177-192) for any twelve residues -- against the oracles, element for element, in ragged batches."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _be(k):
    return k.to_bytes(32, "big")


def test_bls12381_gt_mul_members_nonmembers_and_edge_exponents():
    from kyber_amd.pairing import bls12381 as bls
    from oracle import bls12381 as O

    rng = random.Random(21)
    p = O.P
    member = O.pair(O.g1_mul(rng.randrange(1, O.R), O.G1_GEN), O.g2_mul(rng.randrange(1, O.R), O.G2_GEN))
    x = [(rng.randrange(p), rng.randrange(p)) for _ in range(6)]
    unitary = O.f12_mul(O.f12_frob(x, 6), O.f12_inv(x))
    cyclo = O.f12_mul(O.f12_frob(unitary, 2), unitary)
    outside = [x, [(0, 0)] * 6, unitary, cyclo, O.f12_mul(cyclo, member), [((p - 1), 0)] + [(0, 0)] * 5]
    inside = [member, O.f12_pow(member, 77), O.f12_pow(cyclo, (p ** 4 - p ** 2 + 1) // O.R), O.F12_ONE]
    exps = [0, 1, 2, O.R - 1, O.R, (1 << 256) - 1, rng.getrandbits(256), rng.getrandbits(255) | (1 << 255), rng.getrandbits(128)]
    elems, ks, want, want_st = [], [], [], []
    for a in inside:
        for k in exps:
            elems.append(O.gt_to_bytes(a)); ks.append(_be(k)); want.append(O.gt_to_bytes(O.f12_pow(a, k))); want_st.append(0)
    for a in outside:
        elems.append(O.gt_to_bytes(a)); ks.append(_be(5)); want.append(bytes(576)); want_st.append(2)
    elems.append(b"\xff" * 576); ks.append(_be(5)); want.append(bytes(576)); want_st.append(1)   # a coefficient >= p
    order = list(range(len(elems)))
    rng.shuffle(order)                                             # rejected lanes between accepted ones
    reps = 3                                                       # 3 x 43 = 129 elements: two full batches and one lane
    E = np.frombuffer(b"".join(elems[i] for i in order) * reps, dtype=np.uint8).reshape(-1, 576)
    K = np.frombuffer(b"".join(ks[i] for i in order) * reps, dtype=np.uint8).reshape(-1, 32)
    out, st = bls.gt_batch_mul(K, E)
    assert list(st) == [want_st[i] for i in order] * reps
    for r, i in enumerate(order * reps):
        assert bytes(out[r]) == want[i], (r, i)


@pytest.mark.parametrize("name", ["bn256", "bn254"])
def test_bn_gt_mul_of_any_element(name):
    import importlib

    m = importlib.import_module("kyber_amd.pairing." + name)
    O = importlib.import_module("oracle." + name)
    rng = random.Random(23)
    p = O.P
    strict = name == "bn254"
    gt = O.gt_unmarshal(O.pair_bytes(O.g1_marshal(O.G1_GEN), O.g2_marshal(O.G2_GEN)))
    elems = [gt, [(rng.randrange(p), rng.randrange(p)) for _ in range(6)], [(0, 0)] * 6, O.F12_ONE,
             [(p - 1, p - 1)] * 6]
    exps = [0, 1, O.ORDER - 1, O.ORDER, (1 << 256) - 1, rng.getrandbits(256) | (1 << 255), rng.getrandbits(200)]
    E, K, want, want_st = [], [], [], []
    for a in elems:
        for k in exps:
            E.append(O.gt_marshal(a)); K.append(_be(k)); want.append(O.gt_marshal(O.f12_pow(a, k))); want_st.append(0)
    big = b"\xff" * 384                                             # every coefficient 2^256 - 1
    E.append(big); K.append(_be(3))
    if strict:                                                      # bn254 rejects coefficients >= p (point.go UnmarshalBinary)
        want.append(bytes(384)); want_st.append(1)
    else:                                                           # bn256 reduces them
        want.append(O.gt_mul_bytes(_be(3), big)); want_st.append(0)
    reps = 2                                                        # 72 elements: one full batch and a ragged one
    out, st = m.gt_batch_mul(b"".join(K) * reps, np.frombuffer(b"".join(E) * reps, dtype=np.uint8).reshape(-1, 384))
    assert list(st) == want_st * reps
    for r in range(len(want) * reps):
        assert bytes(out[r]) == want[r % len(want)], r


def test_gt_mul_device_pointers_and_a_second_stream():
    """the _dev entry on CUDA tensors, on a non-default stream, equals the host-buffer call"""
    import torch

    from kyber_amd.pairing import bn256 as bn

    m = 200
    k = np.frombuffer(random.Random(5).randbytes(32 * m), dtype=np.uint8).reshape(m, 32).copy()
    k[:, 0] &= 0x0F
    P, _ = bn.g1_commit(k)
    G2 = np.tile(np.frombuffer(bn.G2_BASE, dtype=np.uint8), (m, 1))
    e, _ = bn.batch_pair(P, G2)
    ref, st = bn.gt_batch_mul(k, e)
    assert not st.any()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        out, st2 = bn.gt_batch_mul(torch.from_numpy(k).cuda(), torch.from_numpy(np.ascontiguousarray(e)).cuda())
    s.synchronize()
    assert not st2.any().item() and (out.cpu().numpy() == ref).all()
