"""Same-base batches through the fixed-base table (kyber_amd/csrc/fixed_base.cuh): share.PriPoly.Commit
(share/poly.go:143-149), key generation, `Point.Mul(s, nil)`.  The table path must return exactly what the
variable-base kernels return for the same (scalar, base) pairs -- compared byte for byte, against each other and,
on a sample, against the oracles -- for the generator (taken from 64 scalars on), an arbitrary base at the batch sizes
that build a table, a repeated base (table reused), a changed base (table rebuilt), and bases UnmarshalBinary rejects."""
import importlib
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SUITES = ["bls12381", "bn256", "bn254"]


def _suite(name):
    import torch

    assert torch.cuda.is_available()
    return importlib.import_module("kyber_amd.pairing." + name), importlib.import_module("oracle." + name)


def _edge_scalars(order, rng, n):
    ks = [0, 1, 2, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, (1 << 250) + 512, (512 << 10) + 513, order - 1, order, order + 1, (1 << 256) - 1, 1 << 255,
          int.from_bytes(b"\x80" * 32, "big"), int.from_bytes(b"\x81" * 32, "big")]
    ks += [rng.randrange(1 << 256) for _ in range(n - len(ks))]
    return np.frombuffer(b"".join(k.to_bytes(32, "big") for k in ks), dtype=np.uint8).reshape(n, 32).copy(), ks


@pytest.mark.parametrize("name", SUITES)
def test_generator_batches_take_the_table_and_match_the_ladder(name):
    """host buffers, 300 scalars times the suite's generators: the table path (generator known: from 64 scalars) against
    the variable-base kernels on the same pairs, and against the oracle on the edge scalars"""
    m, O = _suite(name)
    rng = random.Random(11)
    n = 300
    s, ks = _edge_scalars(m.ORDER, rng, n)
    for grp, base, mul, batch, gen, enc in (
            (1, m.G1_BASE, m.g1_commit, m.g1_batch_mul, O.G1_GEN, getattr(O, "g1_compress", None) or O.g1_marshal),
            (2, m.G2_BASE, m.g2_commit, m.g2_batch_mul, O.G2_GEN, getattr(O, "g2_compress", None) or O.g2_marshal)):
        out, st = mul(s)  # base = None: the generator
        ref, st2 = batch(s, np.frombuffer(base * n, dtype=np.uint8).reshape(n, -1))
        assert not st.any() and not st2.any()
        assert np.array_equal(np.asarray(out), np.asarray(ref))
        omul = O.g1_mul if grp == 1 else O.g2_mul
        for i in list(range(24)) + [n - 1]:
            assert bytes(np.asarray(out)[i]) == enc(omul(ks[i] % m.ORDER, gen)), (name, grp, hex(ks[i]))
        # a second call (table reused) with fewer scalars
        out2, st = mul(s[:100])
        assert not st.any() and np.array_equal(np.asarray(out2), np.asarray(ref)[:100])


@pytest.mark.parametrize("name,grp,n", [("bls12381", 1, 1 << 17), ("bls12381", 2, 1 << 18), ("bn256", 1, 1 << 17),
                                        ("bn256", 2, 1 << 18), ("bn254", 2, 1 << 18)])
def test_arbitrary_base_at_table_sizes_reuse_and_rebuild(name, grp, n):
    """device tensors, the batch sizes from which a table is built for any base: against the variable-base kernels on
    the same pairs; then the same base again (table reused), then another base (table rebuilt)"""
    import torch

    m, O = _suite(name)
    rng = random.Random(12 + grp)
    raw = np.frombuffer(bytes(rng.getrandbits(8) for _ in range(32 * 4096)), dtype=np.uint8).reshape(4096, 32)
    s = torch.from_numpy(np.tile(raw, (n // 4096, 1)).copy()).cuda()
    s[:, 31] ^= torch.arange(n, device="cuda", dtype=torch.int64).to(torch.uint8)  # not all rows equal
    commit = m.g1_commit if grp == 1 else m.g2_commit
    batch = m.g1_batch_mul if grp == 1 else m.g2_batch_mul
    hs = np.frombuffer(b"".join(rng.randrange(1, m.ORDER).to_bytes(32, "big") for _ in range(2)), dtype=np.uint8).reshape(2, 32)
    bases, st = commit(hs, flags=0)  # two arbitrary points of the group (few scalars: the ladder)
    assert not np.asarray(st).any()
    for which in (0, 0, 1):
        b = torch.from_numpy(np.asarray(bases)[which].copy()).cuda()
        out, st = commit(s, b)
        ref, st2 = batch(s, b.repeat(n, 1))
        torch.cuda.synchronize()
        assert not st.any().item() and not st2.any().item()
        assert torch.equal(out, ref), (name, grp, which)
        if which == 1:
            # ... and DIRECTLY against the oracle (not only against the engine's own ladder): 64+ lanes -- first, last,
            # strided -- of the table walk over an arbitrary base at table size
            h = int.from_bytes(bytes(hs[1]), "big")
            omul, gen = (O.g1_mul, O.G1_GEN) if grp == 1 else (O.g2_mul, O.G2_GEN)
            enc = (getattr(O, "g1_compress", None) or O.g1_marshal) if grp == 1 else (getattr(O, "g2_compress", None) or O.g2_marshal)
            base_pt = omul(h, gen)
            lanes = [0, 1, 2, n - 1, n - 2] + list(range(1777, n - 2, n // 60))
            assert len(lanes) >= 64
            got, sc = out[lanes].cpu().numpy(), s[lanes].cpu().numpy()
            for j, i in enumerate(lanes):
                k = int.from_bytes(bytes(sc[j]), "big")
                assert bytes(got[j]) == enc(omul(k % m.ORDER, base_pt)), (name, grp, i)


@pytest.mark.parametrize("name", SUITES)
def test_rejected_and_infinite_bases(name):
    """a base UnmarshalBinary rejects fails every coefficient alike (status, zero output); the point at infinity
    multiplies to itself -- at a batch size that takes the table path"""
    import torch

    m, _ = _suite(name)
    n = 1 << 17
    s = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda")
    bad = torch.zeros(m.G1_LEN, dtype=torch.uint8, device="cuda")
    bad[-1] = 5  # bn: (0, 5) is not on the curve; BLS12-381: compression bit clear
    out, st = m.g1_commit(s, bad)
    torch.cuda.synchronize()
    assert (st != 0).all().item() and not out.any().item()
    inf = torch.from_numpy(np.frombuffer(m.G1_NULL, dtype=np.uint8).copy()).cuda()
    out, st = m.g1_commit(s, inf)
    torch.cuda.synchronize()
    assert not st.any().item() and torch.equal(out, inf.repeat(n, 1))
    # ... and the table of the NEXT valid base is built afresh
    g = torch.from_numpy(np.frombuffer(m.G1_BASE, dtype=np.uint8).copy()).cuda()
    out, st = m.g1_commit(s[:200000], g)
    ref, _ = m.g1_batch_mul(s[:4096], g.repeat(4096, 1))
    torch.cuda.synchronize()
    assert not st.any().item() and torch.equal(out[:4096], ref)


def test_bls12381_flags_through_the_table():
    """uncompressed output, uncompressed + vouched-for input: the table path honours the call's flags"""
    import torch

    m, _ = _suite("bls12381")
    n = 1 << 17
    s = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda")
    h = np.frombuffer((123456789).to_bytes(32, "big"), dtype=np.uint8).reshape(1, 32)
    P, _ = m.g1_commit(h)                                   # compressed
    Pu, _ = m.g1_commit(h, flags=m.F_UNCOMPRESSED_OUT)      # the same point, uncompressed
    Pc = torch.from_numpy(np.asarray(P)[0].copy()).cuda()
    Puc = torch.from_numpy(np.asarray(Pu)[0].copy()).cuda()
    a, st = m.g1_commit(s, Pc)
    b, st2 = m.g1_commit(s, Pc, m.F_UNCOMPRESSED_OUT)
    c, st3 = m.g1_commit(s, Puc, m.F_UNCOMPRESSED | m.F_TRUSTED(0))
    ref, _ = m.g1_batch_mul(s[:8192], Pc.repeat(8192, 1))
    refu, _ = m.g1_batch_mul(s[:8192], Pc.repeat(8192, 1), m.F_UNCOMPRESSED_OUT)
    torch.cuda.synchronize()
    assert not st.any().item() and not st2.any().item() and not st3.any().item()
    assert torch.equal(a[:8192], ref) and torch.equal(b[:8192], refu) and torch.equal(c, a)


def test_bls12381_new_compressed_bases_decoded_with_the_wave():
    """fixed_base.cuh chain_rows_kernel + fb_g1_policy::decode_head / decode_tail: a NEW compressed G1 base has its square
    root's power run on a row.  Both sort flags, an x with no point above it (status 1), a curve point outside the
    subgroup (status 2, read off the table), infinity with a stray flag -- against the per-element ladder / the oracle"""
    import torch

    from oracle import bls12381 as O

    m, _ = _suite("bls12381")
    n = 1 << 16
    s = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda")
    P = O.g1_mul(0xC0FFEE, O.G1_GEN)
    for Q in (P, O.g1_neg(P)):
        base = torch.from_numpy(np.frombuffer(O.g1_compress(Q), dtype=np.uint8).copy()).cuda()
        out, st = m.g1_commit(s, base)
        ref, _ = m.g1_batch_mul(s[:2048], base.repeat(2048, 1))
        torch.cuda.synchronize()
        assert not st.any().item() and torch.equal(out[:2048], ref)
        assert bytes(out[7].cpu().numpy()) == O.g1_mul_bytes(bytes(s[7].cpu().numpy()), O.g1_compress(Q))
    x = 1
    while O.fp_sqrt((x * x * x + 4) % O.P) is not None:
        x += 1
    nox = bytearray(x.to_bytes(48, "big"))
    nox[0] |= 0x80
    out, st = m.g1_commit(s, torch.tensor(list(nox), dtype=torch.uint8, device="cuda"))
    assert (st == 1).all().item() and not out.any().item()
    x = 1
    while True:
        y = O.fp_sqrt((x * x * x + 4) % O.P)
        if y is not None and not O.g1_in_subgroup((x, y)):
            break
        x += 1
    off = torch.from_numpy(np.frombuffer(O.g1_compress((x, y)), dtype=np.uint8).copy()).cuda()
    out, st = m.g1_commit(s, off)
    assert (st == 2).all().item() and not out.any().item()
    bad_inf = torch.tensor([0xE0] + [0] * 47, dtype=torch.uint8, device="cuda")   # infinity with the sort flag
    out, st = m.g1_commit(s, bad_inf)
    assert (st == 1).all().item() and not out.any().item()
    base = torch.from_numpy(np.frombuffer(O.g1_compress(P), dtype=np.uint8).copy()).cuda()  # and a good base afterwards
    out, st = m.g1_commit(s, base)
    assert not st.any().item() and bytes(out[9].cpu().numpy()) == O.g1_mul_bytes(bytes(s[9].cpu().numpy()), O.g1_compress(P))


def test_bls12381_new_g2_bases_through_the_table():
    """a NEW G2 base (fixed_base.cuh chain_kernel on four cooperating lanes): compressed with both sort flags and
    uncompressed, an x with no point (status 1), a curve point outside G2 (status 2, read off the table), a good base
    afterwards -- against the ladder / the oracle"""
    import torch

    from oracle import bls12381 as O

    m, _ = _suite("bls12381")
    n = 1 << 16
    s = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda")
    P = O.g2_mul(0xBADC0DE, O.G2_GEN)
    for Q in (P, O.g2_neg(P)):
        base = torch.from_numpy(np.frombuffer(O.g2_compress(Q), dtype=np.uint8).copy()).cuda()
        out, st = m.g2_commit(s, base)
        ref, _ = m.g2_batch_mul(s[:1024], base.repeat(1024, 1))
        torch.cuda.synchronize()
        assert not st.any().item() and torch.equal(out[:1024], ref)
        assert bytes(out[5].cpu().numpy()) == O.g2_mul_bytes(bytes(s[5].cpu().numpy()), O.g2_compress(Q))
        unc = torch.from_numpy(np.frombuffer(O.g2_serialize_unc(Q), dtype=np.uint8).copy()).cuda()
        out2, st2 = m.g2_commit(s, unc, m.F_UNCOMPRESSED)
        assert not st2.any().item() and torch.equal(out2, out)
    xx = 1
    while O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_sqr((xx, 1)), (xx, 1)), (4, 4))) is not None:
        xx += 1
    nox = bytearray((1).to_bytes(48, "big") + xx.to_bytes(48, "big"))
    nox[0] |= 0x80
    out, st = m.g2_commit(s, torch.tensor(list(nox), dtype=torch.uint8, device="cuda"))
    assert (st == 1).all().item() and not out.any().item()
    xx = 1
    while True:
        c = (xx, 1)
        y = O.f2_sqrt(O.f2_add(O.f2_mul(O.f2_sqr(c), c), (4, 4)))
        if y is not None and not O.g2_in_subgroup((c, y)):
            break
        xx += 1
    off = torch.from_numpy(np.frombuffer(O.g2_compress((c, y)), dtype=np.uint8).copy()).cuda()
    out, st = m.g2_commit(s, off)
    assert (st == 2).all().item() and not out.any().item()
    base = torch.from_numpy(np.frombuffer(O.g2_compress(P), dtype=np.uint8).copy()).cuda()
    out, st = m.g2_commit(s, base)
    assert not st.any().item() and bytes(out[9].cpu().numpy()) == O.g2_mul_bytes(bytes(s[9].cpu().numpy()), O.g2_compress(P))
