"""The lane machine (kyber_amd/csrc/lane_vm.cuh, bls12381_lvm.cuh) on the GPU: (1) the interpreter against the
program simulator of gen_lane_vm.py, record by record, through the trace switch of kyb_debug_bls12381_lvm_trace;
(2) G1Elt.Mul / G2Elt.Mul (kilic/g1.go:110-116, kilic/g2.go) through the public entry points at batch sizes that take the
machine, byte for byte against the oracle: edge scalars, points at infinity, rejected points, results at infinity
(the lanes the per-lane kernel redoes), every input / output form."""
import ctypes as C
import hashlib
import os
import random
import sys

import numpy as np
import pytest

from oracle import bls12381 as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kyber_amd", "csrc"))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bls():
    import torch

    assert torch.cuda.is_available()
    from kyber_amd import _lib
    from kyber_amd.pairing import bls12381 as bls

    # every batch of these tests takes the machine (G1 batches below two waves per SIMD stay on the per-lane kernel by default)
    lib = C.CDLL(_lib.LIB_PATH)
    lib.kyb_debug_bls12381_lvm_min.argtypes = [C.c_longlong]
    assert lib.kyb_debug_bls12381_lvm_min(1024) == 0
    yield bls
    lib.kyb_debug_bls12381_lvm_min(-1)


def _g1_unc(p):
    return p[0].to_bytes(48, "big") + p[1].to_bytes(48, "big")


def _g2_unc(q):
    return q[0][1].to_bytes(48, "big") + q[0][0].to_bytes(48, "big") + q[1][1].to_bytes(48, "big") + q[1][0].to_bytes(48, "big")


@pytest.mark.parametrize("g2", [False, True])
def test_interpreter_follows_the_simulator_record_by_record(bls, g2):
    import torch

    import gen_lane_vm as G
    from kyber_amd import _lib

    lib = C.CDLL(_lib.LIB_PATH)
    fn = lib.kyb_debug_bls12381_lvm_trace
    fn.argtypes = [C.c_int, C.c_size_t] + [C.c_void_p] * 2 + [C.c_size_t] + [C.c_void_p] * 2 + [C.c_uint32, C.c_void_p, C.c_void_p]
    P = G.build_bls12381_g2_mul() if g2 else G.build_bls12381_g1_mul()
    rng = random.Random(9)
    n = 1024
    k0 = rng.randrange(O.R)
    h0 = rng.randrange(1, O.R)
    if g2:
        pt = O.g2_mul(h0, O.G2_GEN)
        wire, ln = _g2_unc(pt), 192
        inputs = [[pt[0][0], pt[1][0]], [pt[0][1], pt[1][1]]]
        digits = [G.bls_g2_digits(k0)] * 2
    else:
        pt = O.g1_mul(h0, O.G1_GEN)
        wire, ln = _g1_unc(pt), 96
        inputs = [[pt[0], pt[1]]]
        digits = [G.bls_g1_digits(k0)]
    trace = []
    outs, flags, _ = P.simulate(inputs, digits, trace=trace)
    pts = torch.from_numpy(np.frombuffer(wire * n, dtype=np.uint8).copy()).cuda()
    ks = torch.from_numpy(np.frombuffer(k0.to_bytes(32, "big") * n, dtype=np.uint8).copy()).cuda()
    out = torch.zeros(n * ln, dtype=torch.uint8, device="cuda")
    st = torch.zeros(n, dtype=torch.uint8, device="cuda")
    tr = torch.zeros((len(trace) + 8, 2, 16), dtype=torch.int32, device="cuda")
    fl = bls.F_UNCOMPRESSED | bls.F_TRUSTED(0) | bls.F_UNCOMPRESSED_OUT
    rc = fn(int(g2), n, ks.data_ptr(), pts.data_ptr(), ln, out.data_ptr(), st.data_ptr(), fl, tr.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    got = tr.cpu().numpy()
    N = P.f.N
    for j, (idx, lanes) in enumerate(trace):
        for lane, limbs in enumerate(lanes):
            dev = [int(x) for x in got[j, lane, :N]]
            assert dev == limbs, "record %d (%s, executed #%d), lane %d: device %s simulator %s" % (idx, P.names[idx], j, lane, dev[:3], limbs[:3])
    exp = O.g2_mul(k0, pt) if g2 else O.g1_mul(k0, pt)
    res = bytes(out.cpu().numpy()[:ln])
    assert res == (_g2_unc(exp) if g2 else _g1_unc(exp))


def _scalars(label, n):
    raw = hashlib.shake_256(label).digest(n * 32)
    a = np.frombuffer(raw, dtype=np.uint8).reshape(n, 32).copy()
    return a


@pytest.mark.parametrize("grp", [1, 2])
def test_mul_through_the_machine_against_the_oracle(bls, grp):
    """n = 3000 elements (above the machine's threshold, not a multiple of 64): random 256-bit scalars (most of them
    above r), edge scalars, infinity and rejected points sprinkled in, every flag combination."""
    n = 3000
    rng = random.Random(grp)
    k = _scalars(b"lvm/k%d" % grp, n)
    edge = [0, 1, 2, 3, O.R - 1, O.R, O.R + 1, 2 * O.R, (1 << 256) - 1, 1 << 255, 0xD201000000010000, 0xD201000000010000 ** 2]
    for j, e in enumerate(edge):
        k[10 + j] = np.frombuffer(e.to_bytes(32, "big"), dtype=np.uint8)
        k[n - 1 - j] = np.frombuffer(e.to_bytes(32, "big"), dtype=np.uint8)
    hs = [rng.randrange(1, O.R) for _ in range(8)]
    if grp == 1:
        base = [O.g1_mul(h, O.G1_GEN) for h in hs]
        comp, unc, mul, ln, dec = O.g1_compress, O.g1_serialize_unc, O.g1_mul, 48, bls.g1_batch_mul
    else:
        base = [O.g2_mul(h, O.G2_GEN) for h in hs]
        comp, unc, mul, ln, dec = O.g2_compress, O.g2_serialize_unc, O.g2_mul, 96, bls.g2_batch_mul
    which = [i % 8 for i in range(n)]
    pts_c = [comp(base[w]) for w in which]
    pts_u = [unc(base[w]) for w in which]
    inf_rows, bad_rows = [5, 100, n - 30], [7, 101, n - 31]
    for i in inf_rows:
        pts_c[i], pts_u[i] = comp(None), unc(None)
    for i in bad_rows:
        pts_c[i] = bytes([0x80]) + bytes(ln - 2) + b"\x05" if grp == 1 else bytes([0x80]) + bytes(ln - 2) + b"\x07"
        pts_u[i] = bytes(2 * ln - 1) + b"\x01"   # (0, 1): not on the curve
    samples = sorted(set(list(range(0, 40)) + list(range(n - 40, n)) + [rng.randrange(n) for _ in range(24)] + inf_rows + bad_rows))
    ki = [int.from_bytes(bytes(k[i]), "big") for i in range(n)]
    for fin, fout in ((0, 0), (bls.F_UNCOMPRESSED, 0), (bls.F_UNCOMPRESSED | bls.F_TRUSTED(0), bls.F_UNCOMPRESSED_OUT), (bls.F_TRUSTED(0), 0)):
        pts = pts_u if fin & bls.F_UNCOMPRESSED else pts_c
        out, st = dec(k, b"".join(pts), fin | fout)
        out, st = np.asarray(out), np.asarray(st)
        enc = unc if fout else comp
        for i in samples:
            if i in bad_rows:
                if fin & bls.F_TRUSTED(0) and fin & bls.F_UNCOMPRESSED:
                    continue  # vouched-for garbage: no contract
                if fin & bls.F_TRUSTED(0) and not (fin & bls.F_UNCOMPRESSED):
                    # compressed, trusted: the square root still has to exist
                    try:
                        (O.g1_decompress if grp == 1 else O.g2_decompress)(pts[i], subgroup_check=False)
                        continue
                    except O.DecodeError:
                        pass
                assert st[i] != 0 and not out[i].any(), (fin, i)
                continue
            assert st[i] == 0, (fin, fout, i, st[i])
            p = None if i in inf_rows else base[which[i]]
            assert bytes(out[i]) == enc(mul(ki[i], p)), (fin, fout, i, hex(ki[i]))
    # a shared base (PriPoly.Commit's shape)
    out, st = (bls.g1_commit if grp == 1 else bls.g2_commit)(k, comp(base[3]))
    out = np.asarray(out)
    for i in samples[::3]:
        assert bytes(out[i]) == comp(mul(ki[i], base[3])), i


@pytest.mark.parametrize("grp,n", [(1, (1 << 19) + 1000), (2, (1 << 18) + 777)])
def test_batches_larger_than_one_chunk(bls, grp, n):
    """The machine works through a large batch in chunks (2^19 G1 / 2^18 G2 elements: the window tables stay bounded):
    elements either side of the chunk boundary, the ragged last chunk, a rejected and an infinite point in the second
    chunk, status bytes of the whole batch -- against the oracle, and the identity sum_i k_i P == (sum k_i) P over ALL
    outputs through the MSM (so that no element of either chunk can be wrong unnoticed)."""
    import torch

    chunk = (1 << 19) if grp == 1 else (1 << 18)
    k = _scalars(b"lvm/chunk/%d" % grp, n)
    k[:, 0] &= 0x3F
    h = 0xC0FFEE
    if grp == 1:
        base, comp, mul, mul_fn, msm, commit = O.g1_mul(h, O.G1_GEN), O.g1_compress, O.g1_mul, bls.g1_batch_mul, bls.g1_msm, bls.g1_commit
    else:
        base, comp, mul, mul_fn, msm, commit = O.g2_mul(h, O.G2_GEN), O.g2_compress, O.g2_mul, bls.g2_batch_mul, bls.g2_msm, bls.g2_commit
    enc = np.frombuffer(comp(base), dtype=np.uint8)
    pts = np.tile(enc, (n, 1)).copy()
    bad, inf = chunk + 5, chunk + 9
    pts[bad] = 0
    pts[bad, 0] = 0x80
    pts[bad, -1] = 5 if grp == 1 else 7
    pts[inf] = np.frombuffer(comp(None), dtype=np.uint8)
    dk, dp = torch.from_numpy(k).cuda(), torch.from_numpy(pts).cuda()
    out, st = mul_fn(dk, dp)
    out_h, st_h = out.cpu().numpy(), st.cpu().numpy()
    assert st_h[bad] != 0 and not np.delete(st_h, bad).any() and not out_h[bad].any()
    assert bytes(out_h[inf]) == comp(None)
    for i in (0, 1, chunk - 2, chunk - 1, chunk, chunk + 1, n - 2, n - 1):
        assert bytes(out_h[i]) == comp(mul(int.from_bytes(bytes(k[i]), "big"), base)), i
    # every output at once: sum of the outputs (unit-scalar MSM) == (sum of the scalars) * base
    keep = np.ones(n, dtype=bool)
    keep[[bad, inf]] = False
    ones = torch.zeros((int(keep.sum()), 32), dtype=torch.uint8, device="cuda")
    ones[:, 31] = 1
    tot, st2 = msm(ones, out[torch.from_numpy(keep).cuda()], bls.F_TRUSTED(0))
    ksum = sum(int.from_bytes(bytes(x), "big") for x in k[keep]) % O.R
    exp, _ = commit((ksum * h % O.R).to_bytes(32, "big"))
    assert not st2.any().item() and bytes(tot.cpu().numpy()) == bytes(np.asarray(exp)[0])


SEEDS = list(range(int(os.environ.get("KYB_SOAK_SEEDS", "2"))))


@pytest.mark.parametrize("seed", SEEDS)
def test_lane_machine_soak(bls, seed):
    """Fresh inputs every seed (KYB_SOAK_SEEDS=n for more): random batch size above the threshold, group, calling
    convention, base points, scalars with edge values at random places, infinite and rejected points at random rows;
    two dozen rows element for element against the oracle and ALL rows through sum_i out_i == sum_i k_i P_i (an MSM over
    the inputs against a unit-scalar MSM over the outputs)."""
    import torch

    rng = random.Random(1000 + seed)
    grp = rng.choice((1, 2))
    n = rng.randrange(1024, 6000)
    fin = rng.choice((0, bls.F_TRUSTED(0), bls.F_UNCOMPRESSED, bls.F_UNCOMPRESSED | bls.F_TRUSTED(0)))
    fout = rng.choice((0, bls.F_UNCOMPRESSED_OUT))
    if grp == 1:
        gen, comp, unc, mul, mul_fn, msm = O.G1_GEN, O.g1_compress, O.g1_serialize_unc, O.g1_mul, bls.g1_batch_mul, bls.g1_msm
    else:
        gen, comp, unc, mul, mul_fn, msm = O.G2_GEN, O.g2_compress, O.g2_serialize_unc, O.g2_mul, bls.g2_batch_mul, bls.g2_msm
    bases = [mul(rng.randrange(1, O.R), gen) for _ in range(6)]
    which = [rng.randrange(6) for _ in range(n)]
    enc_in = unc if fin & bls.F_UNCOMPRESSED else comp
    tab = [enc_in(b) for b in bases]
    pts = [tab[w] for w in which]
    k = np.frombuffer(rng.randbytes(32 * n), dtype=np.uint8).reshape(n, 32).copy()
    for e in (0, 1, 2, O.R - 1, O.R, O.R + 1, (1 << 256) - 1, 0xD201000000010000, (0xD201000000010000 ** 2) - 1):
        k[rng.randrange(n)] = np.frombuffer(e.to_bytes(32, "big"), dtype=np.uint8)
    inf_rows = {rng.randrange(n) for _ in range(3)}
    for i in inf_rows:
        pts[i] = enc_in(None)
    bad_rows = set()
    if not (fin & bls.F_TRUSTED(0)):          # (vouched-for garbage has no contract)
        bad_rows = {rng.randrange(n) for _ in range(3)} - inf_rows
        for i in bad_rows:
            b = bytearray(pts[i])
            b[-1] ^= 1                          # y (or x) off by one: not on the curve / not the encoded point
            if not (fin & bls.F_UNCOMPRESSED):
                b = bytearray(len(b))
                b[0] = 0x80
                b[-1] = 5 if grp == 1 else 7
            pts[i] = bytes(b)
    dk = torch.from_numpy(k).cuda()
    dp = torch.from_numpy(np.frombuffer(b"".join(pts), dtype=np.uint8).copy()).cuda()
    out, st = mul_fn(dk, dp, fin | fout)
    out_h, st_h = out.cpu().numpy(), st.cpu().numpy()
    enc_out = unc if fout else comp
    rows = set(rng.sample(range(n), 20)) | inf_rows | bad_rows | {0, n - 1}
    really_bad = set()
    for i in sorted(rows):
        if i in bad_rows:
            try:  # the mutation may by chance be another valid encoding
                (O.g1_deserialize_unc if grp == 1 else O.g2_deserialize_unc)(pts[i]) if fin & bls.F_UNCOMPRESSED else \
                    (O.g1_decompress if grp == 1 else O.g2_decompress)(pts[i])
            except O.DecodeError:
                really_bad.add(i)
                assert st_h[i] != 0 and not out_h[i].any(), (seed, i)
                continue
        if i in bad_rows:
            continue
        assert st_h[i] == 0, (seed, i, st_h[i])
        p = None if i in inf_rows else bases[which[i]]
        assert bytes(out_h[i]) == enc_out(mul(int.from_bytes(bytes(k[i]), "big"), p)), (seed, grp, hex(fin), hex(fout), i)
    keep = np.ones(n, dtype=bool)
    keep[list(bad_rows | inf_rows)] = False
    dkeep = torch.from_numpy(keep).cuda()
    lhs, s1 = msm(dk[dkeep], dp.view(n, -1)[dkeep], fin | bls.F_TRUSTED(0))
    ones = torch.zeros((int(keep.sum()), 32), dtype=torch.uint8, device="cuda")
    ones[:, 31] = 1
    rhs, s2 = msm(ones, out.view(n, -1)[dkeep], bls.F_TRUSTED(0) | (bls.F_UNCOMPRESSED if fout else 0))
    assert not s1.any().item() and not s2.any().item() and bytes(lhs.cpu().numpy()) == bytes(rhs.cpu().numpy()), seed


def test_lane_machine_from_several_threads_and_streams(bls):
    """Two host threads through the host-buffer entry points and two device streams at once: every call has its own
    (kind, stream) workspace -- tables, digits, redo mask -- and returns what it returned alone."""
    import threading

    import torch

    n = 3000
    k = _scalars(b"lvm/thr/k", n)
    q = np.frombuffer(O.g2_compress(O.g2_mul(0xABCDEF, O.G2_GEN)) * n, dtype=np.uint8).reshape(n, 96).copy()
    p = np.frombuffer(O.g1_compress(O.g1_mul(0xFEDCBA, O.G1_GEN)) * n, dtype=np.uint8).reshape(n, 48).copy()
    ref2 = np.asarray(bls.g2_batch_mul(k, q)[0]).copy()
    ref1 = np.asarray(bls.g1_batch_mul(k, p)[0]).copy()
    errors = []

    def host(fn, arr, ref):
        try:
            for _ in range(4):
                if not (np.asarray(fn(k, arr)[0]) == ref).all():
                    errors.append("host mismatch")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def dev(fn, arr, ref):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                dk, da = torch.from_numpy(k).cuda(), torch.from_numpy(arr).cuda()
                for _ in range(4):
                    out, _ = fn(dk, da)
                    st.synchronize()
                    if not (out.cpu().numpy() == ref).all():
                        errors.append("device mismatch")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=host, args=(bls.g2_batch_mul, q, ref2), daemon=True),
               threading.Thread(target=host, args=(bls.g1_batch_mul, p, ref1), daemon=True),
               threading.Thread(target=dev, args=(bls.g2_batch_mul, q, ref2), daemon=True),
               threading.Thread(target=dev, args=(bls.g1_batch_mul, p, ref1), daemon=True)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "threads are stuck"
    assert not errors, errors
