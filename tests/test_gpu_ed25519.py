"""GPU parity tests for the Ed25519 hot path, through the C ABI, bit-exact
against the oracle and the reference's golden vectors."""
import hashlib
import os

import numpy as np
import pytest

from oracle import ed25519 as O
from tests import _oracle_c as OC

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
KAT = np.load(os.path.join(G, "ed25519_sign_input.npy"))


@pytest.fixture(scope="module")
def ed():
    import torch

    assert torch.cuda.is_available()
    from kyber_amd.group import edwards25519 as ed

    return ed


def test_base_table_built_on_device_is_correct(ed):
    from kyber_amd import _lib

    tab = np.zeros(33 * 136 * 32, dtype=np.int32)  # (pos, |digit| - 1, 30 limbs + 2 pad words)
    _lib.check(_lib.load().kyb_ed25519_debug_base_table(tab.ctypes.data), "table")
    tab = tab.reshape(33, 136, 32)[:, :, :30].reshape(33, 136, 3, 10)

    def val(l):
        x, off = 0, 0
        for i in range(10):
            x += int(l[i]) << off
            off += 25 if i & 1 else 26
        return x % O.P

    for pos in (0, 1, 7, 31, 32):
        for j in (0, 1, 7, 8, 15, 16, 127, 128, 135):
            x, y = O.mul_int((j + 1) << (8 * pos), O.B)
            assert val(tab[pos, j, 0]) == (y + x) % O.P
            assert val(tab[pos, j, 1]) == (y - x) % O.P
            assert val(tab[pos, j, 2]) == 2 * O.D * x * y % O.P


def test_fixed_base_golden_1024(ed):
    # sign.input: pub = a*B (clamped, unreduced a), R = r*B
    assert (ed.batch_mul_base(KAT[:, 0]) == KAT[:, 1]).all()
    assert (ed.batch_mul_base(KAT[:, 2]) == KAT[:, 3]).all()


def test_var_base_golden_verify_equation(ed):
    # h*A from the engine must satisfy S*B = R + h*A (eddsa.go:219-227)
    hA, st = ed.batch_mul(KAT[:, 4], KAT[:, 1])
    assert not st.any()
    SB = ed.batch_mul_base(KAT[:, 5])
    for i in range(0, len(KAT), 4):
        assert O.add(O.decode(bytes(KAT[i, 3])), O.decode(bytes(hA[i]))) == O.decode(bytes(SB[i]))


@pytest.mark.parametrize("vartime", [False, True])
def test_var_base_random_vs_c_oracle(ed, vartime):
    rng = np.random.default_rng(11)
    n = 8192 + 37  # ragged: not a multiple of the block size
    scalars = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)  # any 256-bit value
    pts = OC.ed_mul_base(rng.integers(0, 256, size=(n, 32), dtype=np.uint8))
    exp, est = OC.ed_mul(scalars, pts, vartime=vartime)
    out, st = ed.batch_mul(scalars, pts, vartime=vartime)
    assert (st == est).all() and (out == exp).all()


@pytest.mark.parametrize("vartime", [False, True])
def test_fixed_base_random_vs_c_oracle(ed, vartime):
    rng = np.random.default_rng(12)
    n = 10000 + 3
    scalars = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    out = ed.batch_mul_base(scalars, vartime=vartime)
    if vartime:
        # full 256-bit semantics: compare with variable-base-vartime oracle on B
        exp, _ = OC.ed_mul(scalars, np.tile(np.frombuffer(O.encode(O.B), dtype=np.uint8), (n, 1)), vartime=True)
    else:
        exp = OC.ed_mul_base(scalars)
    assert (out == exp).all()


def test_edge_cases_vs_python_oracle(ed):
    enc_b = O.encode(O.B)
    scalars = [bytes(32), (1).to_bytes(32, "little"), O.L.to_bytes(32, "little"),
               (O.L - 1).to_bytes(32, "little"), bytes([0xFF] * 32), (2**255).to_bytes(32, "little"),
               (2**255 - 1).to_bytes(32, "little"), (8).to_bytes(32, "little"),
               bytes([0x88] * 32), bytes([0x08] * 32), (2**252).to_bytes(32, "little")]
    import json
    misc = json.load(open(os.path.join(G, "ed25519_misc.json")))
    points = [enc_b, b"\x01" + bytes(31)] + [bytes.fromhex(h) for h in misc["small_order"]]
    # non-canonical encodings (y >= p) and "-0": accepted by the reference decode
    points += [(O.P + 1).to_bytes(32, "little"), ((O.P + 1) | (1 << 255)).to_bytes(32, "little"),
               (1 | (1 << 255)).to_bytes(32, "little"), (2).to_bytes(32, "little"),  # y=2: not on curve
               bytes(KAT[3, 1])]
    S, Pn = [], []
    for s in scalars:
        for p in points:
            S.append(s)
            Pn.append(p)
    S = np.frombuffer(b"".join(S), dtype=np.uint8).reshape(-1, 32)
    Pn = np.frombuffer(b"".join(Pn), dtype=np.uint8).reshape(-1, 32)
    for vt in (False, True):
        out, st = ed.batch_mul(S, Pn, vartime=vt)
        for i in range(len(S)):
            exp = O.mul(bytes(S[i]), bytes(Pn[i]), vartime=vt)
            if exp is None:
                assert st[i] == 1 and not out[i].any()
            else:
                assert st[i] == 0 and bytes(out[i]) == exp, (i, vt)
    outb = ed.batch_mul_base(np.frombuffer(b"".join(scalars), dtype=np.uint8).reshape(-1, 32))
    for i, s in enumerate(scalars):
        assert bytes(outb[i]) == O.mul_base(s)


def test_empty_and_single(ed):
    out = ed.batch_mul_base(np.zeros((0, 32), dtype=np.uint8))
    assert out.shape == (0, 32)
    out, st = ed.batch_mul(np.zeros((0, 32), dtype=np.uint8), np.zeros((0, 32), dtype=np.uint8))
    assert out.shape == (0, 32) and st.shape == (0,)
    one = ed.batch_mul_base((5).to_bytes(32, "little"))
    assert bytes(one[0]) == O.mul_base((5).to_bytes(32, "little"))


def test_commit_same_base(ed):
    rng = np.random.default_rng(5)
    coeffs = rng.integers(0, 256, size=(300, 32), dtype=np.uint8)
    coeffs[:, 31] &= 0x0F
    base = bytes(KAT[9, 1])
    got = ed.commit(coeffs, base)
    exp, _ = OC.ed_mul(coeffs, np.tile(np.frombuffer(base, dtype=np.uint8), (300, 1)))
    assert (got == exp).all()
    assert (ed.commit(coeffs) == OC.ed_mul_base(coeffs)).all()
    with pytest.raises(ValueError):
        ed.commit(coeffs, (2).to_bytes(32, "little"))


def test_full_size_properties_2p20(ed):
    """BASELINE config 2 size.  Size-independent properties, device-resident:
    a*(b*B) == (a*b mod l)*B and a SHA-256 checksum of a slice against the C oracle."""
    import torch

    n = 1 << 20
    rng = np.random.default_rng(2024)
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    b = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    a[:, 31] &= 0x0F  # < 2^252 < l : canonical
    b[:, 31] &= 0x0F
    d_a, d_b = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    d_bB = ed.batch_mul_base(d_b)
    d_abB, d_st = ed.batch_mul(d_a, d_bB)
    torch.cuda.synchronize()
    assert int(d_st.sum().item()) == 0
    # host: ab mod l for a strided sample (python ints), then fixed-base on the engine
    idx = np.arange(0, n, 257)
    ab = np.zeros((len(idx), 32), dtype=np.uint8)
    for k, i in enumerate(idx):
        v = int.from_bytes(bytes(a[i]), "little") * int.from_bytes(bytes(b[i]), "little") % O.L
        ab[k] = np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint8)
    exp = ed.batch_mul_base(ab)
    got = d_abB.cpu().numpy()
    assert (got[idx] == exp).all()
    # checksum of a contiguous slice vs the C oracle
    sl = slice(n - 4096, n)
    ref, _ = OC.ed_mul(a[sl], d_bB.cpu().numpy()[sl])
    assert hashlib.sha256(got[sl].tobytes()).digest() == hashlib.sha256(ref.tobytes()).digest()


def test_point_scalar_mirror_api(ed):
    suite = ed.NewSuite()
    s = suite.Scalar().SetInt64(7)
    P = suite.Point().Mul(s, None)
    assert P.MarshalBinary() == O.mul_base((7).to_bytes(32, "little"))
    Q = suite.Point().Mul(suite.Scalar().SetInt64(3), P)
    assert Q.MarshalBinary() == O.mul_base((21).to_bytes(32, "little"))
    assert Q.Equal(suite.Point().Mul(suite.Scalar().SetInt64(21), None))
    with pytest.raises(ValueError):
        suite.Point().UnmarshalBinary((2).to_bytes(32, "little"))
    with pytest.raises(TypeError):
        suite.Point().Mul(b"notascalar", None)
    # Mul by zero gives the identity encoding (util/test/test.go:364)
    assert suite.Point().Mul(suite.Scalar().Zero(), P).MarshalBinary() == b"\x01" + bytes(31)
    # RFC 8032 key derivation through NewKeyAndSeedWithInput (curve.go:51-60)
    import json
    misc = json.load(open(os.path.join(G, "ed25519_misc.json")))
    for v in misc["rfc8032"]:
        sec, _, _ = suite.NewKeyAndSeedWithInput(bytes.fromhex(v["seed"]))
        assert suite.Point().Mul(sec, None).MarshalBinary().hex() == v["pub"]


def test_hash_to_curve_rfc9380_vectors(ed):
    """point_test.go:405-445 TestHashToPoint through the engine, plus batch vs oracle."""
    import json

    M = json.load(open(os.path.join(G, "ed25519_misc.json")))
    dst = M["rfc9380_dst"]
    for v in M["rfc9380"]:
        P = ed.Point().Hash(v["msg"].encode(), dst)
        assert O.decode(P.MarshalBinary()) == (int(v["x"], 16), int(v["y"], 16))
    msgs = [hashlib.sha256(b"h%d" % i).digest() for i in range(300)]
    out = ed.batch_hash(msgs, b"kyber-test-DST")
    for i in range(0, 300, 23):
        assert bytes(out[i]) == O.hash_to_curve(msgs[i], b"kyber-test-DST")
    with pytest.raises(Exception):
        ed.batch_hash(msgs[:1], b"")  # the reference rejects an empty domain separator (point.go:365)


def test_two_streams_do_not_share_workspaces(ed):
    """Large batches park window tables / projective results (and MSMs their pipeline arrays) in per-(device, stream)
    workspaces: the same calls issued back to back on two streams, without synchronising in between, must give what
    they give one after the other."""
    import torch

    n = 1 << 14
    rng = np.random.default_rng(11)
    sa = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    sb = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    sa[:, 31] &= 0x7F
    sb[:, 31] &= 0x7F
    ha = sa.copy()
    ha[:, 31] &= 0x0F
    dsa, dsb = torch.from_numpy(sa).cuda(), torch.from_numpy(sb).cuda()
    pts = ed.batch_mul_base(torch.from_numpy(ha).cuda())
    ref_a, _ = ed.batch_mul(dsa, pts)
    ref_b, _ = ed.batch_mul(dsb, pts)
    ref_ma, _ = ed.msm(dsa, pts)
    ref_mb, _ = ed.msm(dsb, pts)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(s1):
            out_a, _ = ed.batch_mul(dsa, pts)
            m_a, _ = ed.msm(dsa, pts)
        with torch.cuda.stream(s2):
            out_b, _ = ed.batch_mul(dsb, pts)
            m_b, _ = ed.msm(dsb, pts)
        torch.cuda.synchronize()
        assert torch.equal(out_a, ref_a) and torch.equal(out_b, ref_b)
        assert torch.equal(m_a, ref_ma) and torch.equal(m_b, ref_mb)


def test_batch_unmarshal_vs_oracle(ed):
    """kyb_ed25519_unmarshal = N x UnmarshalBinary + MarshalBinary (point.go:54-70, ge.go:99-150): non-canonical y and
    "-0" are accepted and come back canonical, a y with no x is status 1, host and device entry points agree."""
    import json
    import torch

    misc = json.load(open(os.path.join(G, "ed25519_misc.json")))
    pts = [O.encode(O.B), b"\x01" + bytes(31)] + [bytes.fromhex(h) for h in misc["small_order"]]
    pts += [(O.P + 1).to_bytes(32, "little"), ((O.P + 1) | (1 << 255)).to_bytes(32, "little"),
            (1 | (1 << 255)).to_bytes(32, "little"), (2).to_bytes(32, "little"), bytes([0xFF] * 32)]
    pts += [bytes(KAT[i, 1]) for i in range(200)]
    raw = hashlib.shake_256(b"ed25519-unmarshal").digest(32 * 300)
    pts += [raw[32 * i:32 * i + 32] for i in range(300)]  # about half of random strings decode
    out, st = ed.batch_unmarshal(b"".join(pts))
    bad = 0
    for i, p in enumerate(pts):
        dec = O.decode(p)
        if dec is None:
            bad += 1
            assert st[i] == 1 and not out[i].any(), i
        else:
            assert st[i] == 0 and bytes(out[i]) == O.encode(dec), i
    assert 100 < bad < 250
    t = torch.frombuffer(bytearray(b"".join(pts)), dtype=torch.uint8).cuda()
    out_d, st_d = ed.batch_unmarshal(t)
    assert out_d.cpu().numpy().tobytes() == out.tobytes() and st_d.cpu().numpy().tobytes() == st.tobytes()


def test_stream_release_frees_and_the_stream_stays_usable(ed):
    """kyb_stream_release: the workspaces tied to a stream handle go away (after the stream drains); the next call on
    the same stream simply allocates again, and releasing a stream that never had one is a no-op."""
    import torch

    import kyber_amd

    n = 1 << 13
    rng = np.random.default_rng(3)
    s = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    s[:, 31] &= 0x0F
    d = torch.from_numpy(s).cuda()
    pts = ed.batch_mul_base(d)
    ref, _ = ed.msm(d, pts)
    st = torch.cuda.Stream()
    kyber_amd.release_stream(st)  # nothing allocated yet
    for _ in range(2):
        with torch.cuda.stream(st):
            out, _ = ed.msm(d, pts)
            mul, _ = ed.batch_mul(d, pts)
        kyber_amd.release_stream(st)  # waits for the stream itself
        assert torch.equal(out, ref)
    with torch.cuda.stream(st):
        out, _ = ed.msm(d, pts)
    st.synchronize()
    before = torch.cuda.mem_get_info()[0]
    kyber_amd.release_stream(st)
    assert torch.cuda.mem_get_info()[0] >= before  # the MSM workspace went back to the driver
    assert torch.equal(out, ref)


def test_same_base_commit_through_its_own_table(ed):
    """kyb_ed25519_mul_same_base with >= 16384 coefficients builds a radix-256 table for the shared base and takes the
    fixed-base path (share.PriPoly.Commit with b != nil, share/poly.go:143-149): same bytes as the per-element ladder,
    including scalars >= 2^255, the var-time flag, and a base that does not decode."""
    import hashlib

    import torch

    n = 20000
    s = np.frombuffer(hashlib.shake_256(b"commit/s").digest(n * 32), dtype=np.uint8).reshape(n, 32).copy()  # all 256 bits
    base = ed.batch_mul_base(np.frombuffer(hashlib.sha256(b"commit/base").digest(), dtype=np.uint8).reshape(1, 32) & 0x7F)[0]
    tiled = torch.from_numpy(np.tile(base, (n, 1))).cuda()
    for vt in (False, True):
        got = ed.commit(s, bytes(base), vartime=vt)
        ref, st = ed.batch_mul(torch.from_numpy(s).cuda(), tiled, vartime=vt)
        assert not st.any().item() and (got == ref.cpu().numpy()).all(), vt
    for i in (0, 1, n - 1):
        assert bytes(got[i]) == O.mul(bytes(s[i]), bytes(base), vartime=True)
    small = ed.commit(s[:100], bytes(base))  # below the threshold: the ladder
    assert (small == ed.commit(s, bytes(base))[:100]).all()
    bad = bytearray(32)
    bad[0] = 2  # y = 2 is not on the curve
    with pytest.raises(ValueError):
        ed.commit(s, bytes(bad))


def test_uniform_access_variants_give_the_default_bytes(ed):
    """KYB_F_UNIFORM (table scanned with masks, no digit-indexed load, no skipped window: the access pattern of the
    reference's constant-time Mul, group/edwards25519/ge.go:352-371, 419-435) changes addresses, not values: fixed-base,
    variable-base (scratch-table and global-table kernels) and same-base batches against the C oracle, over any 256-bit
    scalar -- the >= 2^255 behaviour of the constant-time path included -- from host and device buffers; a base that does
    not decode; and the flag combination that makes no sense is rejected."""
    import torch

    from kyber_amd import _lib

    rng = np.random.default_rng(61)
    edge = [bytes(32), (1).to_bytes(32, "little"), O.L.to_bytes(32, "little"), bytes([0xFF] * 32), (2**255).to_bytes(32, "little"),
            (2**255 - 1).to_bytes(32, "little"), bytes([0x88] * 32), bytes([0x08] * 32), (2**252).to_bytes(32, "little"),
            bytes([0x99] * 32), bytes([0x77] * 31 + [0x8F])]
    for n in (1, 300, 4096 + 77, 20000):
        s = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        for j, e in enumerate(edge[:n]):
            s[(j * 37) % n] = np.frombuffer(e, dtype=np.uint8)
        # fixed base
        exp = OC.ed_mul_base(s)
        assert (ed.batch_mul_base(s, uniform=True) == exp).all(), n
        assert (ed.batch_mul_base(torch.from_numpy(s).cuda(), uniform=True).cpu().numpy() == exp).all(), n
        # variable base (n < 4096: the table in scratch; above: the global slab), one undecodable point
        pts = OC.ed_mul_base(rng.integers(0, 256, size=(n, 32), dtype=np.uint8))
        if n > 2:
            pts[n // 2] = 0
            pts[n // 2, 0] = 2
        exp_v, est = OC.ed_mul(s, pts)
        out, st = ed.batch_mul(s, pts, uniform=True)
        assert (st == est).all() and (out == exp_v).all(), n
        out, st = ed.batch_mul(torch.from_numpy(s).cuda(), torch.from_numpy(pts).cuda(), uniform=True)
        assert (st.cpu().numpy() == est).all() and (out.cpu().numpy() == exp_v).all(), n
        # same base: the table of the base at any batch size
        base = bytes(pts[0])
        exp_c, _ = OC.ed_mul(s, np.tile(pts[0], (n, 1)))
        assert (ed.commit(s, base, uniform=True) == exp_c).all(), n
    bad = bytearray(32)
    bad[0] = 2
    with pytest.raises(ValueError):
        ed.commit(s[:5], bytes(bad), uniform=True)
    with pytest.raises(ValueError):
        ed.batch_mul_base(s[:5], vartime=True, uniform=True)
    lib = _lib.load()
    o = np.zeros((5, 32), dtype=np.uint8)
    assert lib.kyb_ed25519_mul_base(5, s.ctypes.data, o.ctypes.data, _lib.KYB_F_VARTIME | _lib.KYB_F_UNIFORM) == -1
