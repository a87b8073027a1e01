"""Several devices behind ONE host-buffer call (kyb_set_devices): the GPU box has one MI355X, so the device set lists
it three times -- every slice runs the real single-device path on its own host thread and the partition, the per-slice
pointer arithmetic, the status bytes and the MSM's combination of the partial points are all exercised; results must
equal the single-device call byte for byte."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _shake(label, n):
    return np.frombuffer(hashlib.shake_256(label).digest(n), dtype=np.uint8)


@pytest.fixture()
def three_shards():
    from kyber_amd import devices

    devices.set_devices([0, 0, 0])
    devices.set_shard_threshold(1)
    assert devices.get_devices() == [0, 0, 0]
    yield devices
    devices.set_devices([])
    devices.set_shard_threshold(16384)


def _both(devices, fn):
    """fn() with the device set cleared (reference) and with it active"""
    cur = devices.get_devices()
    devices.set_devices([])
    ref = fn()
    devices.set_devices(cur)
    return ref, fn()


def _same(a, b):
    a = a if isinstance(a, tuple) else (a,)
    b = b if isinstance(b, tuple) else (b,)
    return all((np.asarray(x) == np.asarray(y)).all() for x, y in zip(a, b))


def test_ed25519_host_calls_sharded(three_shards):
    from kyber_amd.group import edwards25519 as ed

    n = 1000  # 334 + 333 + 333
    s = _shake(b"md/ed/s", n * 32).reshape(n, 32).copy()
    h = _shake(b"md/ed/h", n * 32).reshape(n, 32).copy()
    s[:, 31] &= 0x7F
    h[:, 31] &= 0x0F
    ref, got = _both(three_shards, lambda: ed.batch_mul_base(h))
    assert _same(ref, got)
    pts = ref.copy()
    for i in (0, 334, 999):  # a rejected point in every slice
        pts[i] = 0
        pts[i, 0] = 2
    ref, got = _both(three_shards, lambda: ed.batch_mul(s, pts))
    assert _same(ref, got) and sorted(np.nonzero(got[1])[0].tolist()) == [0, 334, 999]
    ref, got = _both(three_shards, lambda: ed.commit(s, bytes(pts[5])))
    assert _same(ref, got)
    good = ed.batch_mul_base(h)
    ref, got = _both(three_shards, lambda: ed.msm(s, good))
    assert _same(ref, got) and not np.asarray(got[1]).any()
    ref, got = _both(three_shards, lambda: ed.msm(s, pts))  # rejected points: zero output, statuses in place
    assert _same(ref, got) and not np.asarray(got[0]).any() and np.asarray(got[1]).sum() == 3
    # fewer units than shards, and the threshold: both stay on one device and still agree
    ref, got = _both(three_shards, lambda: ed.batch_mul(s[:2], good[:2]))
    assert _same(ref, got)
    three_shards.set_shard_threshold(5000)
    assert _same(ref, ed.batch_mul(s[:2], good[:2]))


@pytest.mark.parametrize("name", ["bls12381", "bn256"])
def test_pairing_suite_host_calls_sharded(three_shards, name):
    import importlib

    m = importlib.import_module("kyber_amd.pairing." + name)
    n = 200
    k = _shake(b"md/k/" + name.encode(), n * 32).reshape(n, 32).copy()
    a = _shake(b"md/a/" + name.encode(), n * 32).reshape(n, 32).copy()
    k[:, 0] &= 0x3F
    a[:, 0] &= 0x3F
    ref, got = _both(three_shards, lambda: m.g1_commit(a))
    assert _same(ref, got)
    P = np.asarray(got[0])
    Q = np.asarray(m.g2_commit(k)[0])
    ref, got = _both(three_shards, lambda: m.g1_batch_mul(k, P))
    assert _same(ref, got)
    ref, got = _both(three_shards, lambda: m.g2_batch_mul(a, Q))
    assert _same(ref, got)
    Pbad = P.copy()
    Pbad[[3, 70, 199]] = 0xFF
    ref, got = _both(three_shards, lambda: m.batch_pair(Pbad, Q))
    assert _same(ref, got) and sorted(np.nonzero(got[1])[0].tolist()) == [3, 70, 199]
    kP = np.asarray(m.g1_batch_mul(k, P)[0])
    G2 = np.tile(np.frombuffer(m.G2_BASE, dtype=np.uint8), (n, 1))
    forged = kP.copy()
    forged[::7] = P[::7]
    ref, got = _both(three_shards, lambda: m.batch_validate_pairing(P, Q, forged, G2))
    assert _same(ref, got)
    exp = np.ones(n, dtype=bool)
    exp[::7] = False
    assert (np.asarray(got[0]).astype(bool) == exp).all()
    ref, got = _both(three_shards, lambda: m.g1_msm(k, P))
    assert _same(ref, got) and np.asarray(got[0]).any()
    ref, got = _both(three_shards, lambda: m.g2_msm(k, Q))
    assert _same(ref, got)
    ref, got = _both(three_shards, lambda: m.g1_msm(k, Pbad))
    assert _same(ref, got) and not np.asarray(got[0]).any()


def test_bls_verify_sharded(three_shards):
    from kyber_amd.pairing import bls12381 as bls

    n = 96
    x = _shake(b"md/v/x", n * 32).reshape(n, 32).copy()
    x[:, 0] &= 0x3F
    msgs = _shake(b"md/v/m", n * 32).reshape(n, 32).copy()
    X = np.asarray(bls.g2_commit(x)[0])
    Hm = np.asarray(bls.batch_hash_g1(msgs)[0])
    sig = np.asarray(bls.g1_batch_mul(x, Hm)[0]).copy()
    sig[::5] = Hm[::5]
    ref, got = _both(three_shards, lambda: bls.batch_verify_g1(X, msgs, sig))
    assert _same(ref, got)
    exp = np.ones(n, dtype=bool)
    exp[::5] = False
    assert (np.asarray(got[0]).astype(bool) == exp).all()


def test_eight_listed_devices_msm_and_batches():
    """The shape of the 8-GPU node the driver scales to, on the one GPU there is: eight shards (eight host threads, eight
    device contexts' worth of calls on device 0), uneven split (1003 = 3 x 126 + 5 x 125), a rejected point in the
    shard of a middle device, the MSM's combine of eight partial points, and a batch big enough that the BLS12-381 G2
    shards take the lane machine."""
    from kyber_amd import devices
    from kyber_amd.group import edwards25519 as ed
    from kyber_amd.pairing import bls12381 as bls

    devices.set_devices([0] * 8)
    devices.set_shard_threshold(1)
    try:
        assert devices.get_devices() == [0] * 8
        n = 1003
        s = _shake(b"md8/s", n * 32).reshape(n, 32).copy()
        h = _shake(b"md8/h", n * 32).reshape(n, 32).copy()
        s[:, 31] &= 0x7F
        h[:, 31] &= 0x0F
        good = ed.batch_mul_base(h)
        ref, got = _both(devices, lambda: ed.msm(s, good))
        assert _same(ref, got) and np.asarray(got[0]).any()
        bad = good.copy()
        bad[126 * 3 + 125 + 7] = 0
        bad[126 * 3 + 125 + 7, 0] = 2          # inside the fifth shard
        ref, got = _both(devices, lambda: ed.msm(s, bad))
        assert _same(ref, got) and not np.asarray(got[0]).any() and np.asarray(got[1]).sum() == 1
        ref, got = _both(devices, lambda: ed.batch_mul(s, bad))
        assert _same(ref, got) and np.nonzero(got[1])[0].tolist() == [126 * 3 + 125 + 7]
        m = 8 * 1100 + 5                          # every shard above the lane machine's G2 threshold
        k = _shake(b"md8/k", m * 32).reshape(m, 32).copy()
        k[:, 0] &= 0x3F
        Q = np.asarray(bls.g2_commit(k[:64])[0])
        Qs = np.tile(Q, (m // 64 + 1, 1))[:m].copy()
        ref, got = _both(devices, lambda: bls.g2_batch_mul(k, Qs))
        assert _same(ref, got) and not np.asarray(got[1]).any()
        ref, got = _both(devices, lambda: bls.g1_msm(k[:n], np.asarray(bls.g1_commit(k[:n])[0])))
        assert _same(ref, got)
    finally:
        devices.set_devices([])
        devices.set_shard_threshold(16384)


def test_eight_listed_devices_pair_check_verify_and_commit():
    """the eight-shard shape for ValidatePairing, the fused bls.Verify and share.PriPoly.Commit (a
    fixed-base table per device context; here eight shards race for the one context's table) -- uneven split, a forged
    signature / a rejected point in several shards, generator and arbitrary bases, byte for byte against one device."""
    from kyber_amd import devices
    from kyber_amd.pairing import bls12381 as bls, bn256

    devices.set_devices([0] * 8)
    devices.set_shard_threshold(1)
    try:
        n = 8 * 70 + 3
        k = _shake(b"md8p/k", n * 32).reshape(n, 32).copy()
        a = _shake(b"md8p/a", n * 32).reshape(n, 32).copy()
        k[:, 0] &= 0x3F
        a[:, 0] &= 0x3F
        for m in (bls, bn256):
            # Commit: the generator (table known from 64 scalars per shard) and an arbitrary base, twice (table reused)
            ref, got = _both(devices, lambda: m.g1_commit(a))
            assert _same(ref, got) and not np.asarray(got[1]).any()
            base = bytes(np.asarray(got[0])[5])
            for _ in range(2):
                ref, got = _both(devices, lambda: m.g1_commit(k, base))
                assert _same(ref, got) and not np.asarray(got[1]).any()
            ref, got = _both(devices, lambda: m.g2_commit(k))
            assert _same(ref, got)
            P = np.asarray(m.g1_commit(a)[0])
            Q = np.asarray(got[0])
            kP = np.asarray(m.g1_batch_mul(k, P)[0])
            G2 = np.tile(np.frombuffer(m.G2_BASE, dtype=np.uint8), (n, 1))
            forged = kP.copy()
            forged[::9] = P[::9]
            forged[[71, 300, n - 1]] = 0xFF          # undecodable: status, not just false
            ref, got = _both(devices, lambda: m.batch_validate_pairing(P, Q, forged, G2))
            assert _same(ref, got)
            exp = np.ones(n, dtype=bool)
            exp[::9] = False
            exp[[71, 300, n - 1]] = False
            assert (np.asarray(got[0]).astype(bool) == exp).all()
            assert sorted(np.nonzero(np.asarray(got[1]))[0].tolist()) == [71, 300, n - 1]
        msgs = _shake(b"md8p/m", n * 32).reshape(n, 32).copy()
        X = np.asarray(bls.g2_commit(k)[0])
        Hm = np.asarray(bls.batch_hash_g1(msgs)[0])
        sig = np.asarray(bls.g1_batch_mul(k, Hm)[0]).copy()
        sig[::11] = Hm[::11]
        ref, got = _both(devices, lambda: bls.batch_verify_g1(X, msgs, sig))
        assert _same(ref, got)
        exp = np.ones(n, dtype=bool)
        exp[::11] = False
        assert (np.asarray(got[0]).astype(bool) == exp).all()
    finally:
        devices.set_devices([])
        devices.set_shard_threshold(16384)


def test_device_side_combine_of_gathered_partials():
    """kyber_amd/dist.py _tree_sum on CUDA tensors (what the node-wide MSM runs after the RCCL all-gather since round 3:
    batched Point.Add on device tensors, one host read at the end) against the host path, for every suite and for
    counts that make the tree odd at some level; a rejected partial must come back as failure."""
    import torch

    from kyber_amd import dist as kd
    from kyber_amd.group import edwards25519 as ed
    from kyber_amd.pairing import bls12381 as bls, bn256 as bn

    k = _shake(b"tree/k", 8 * 32).reshape(8, 32).copy()
    k[:, 0] &= 0x3F
    for name, pts, add in (("bls g1", np.asarray(bls.g1_commit(k)[0]), lambda a, b: bls.ENGINE.add(1, a, b)),
                           ("bls g2", np.asarray(bls.g2_commit(k)[0]), lambda a, b: bls.ENGINE.add(2, a, b)),
                           ("bn256 g1", np.asarray(bn.g1_commit(k)[0]), lambda a, b: bn.ENGINE.add(1, a, b))):
        for m in (1, 2, 3, 5, 8):
            host, ok_h = kd._tree_sum(pts[:m].copy(), add)
            dev, ok_d = kd._tree_sum(torch.from_numpy(pts[:m].copy()).cuda(), add)
            assert ok_h and ok_d and bytes(dev.cpu().numpy()) == bytes(np.asarray(host)), (name, m)
    s = k.copy()
    s[:, 31] &= 0x0F
    e = np.asarray(ed.batch_mul_base(s[:, ::-1].copy()))
    for m in (1, 4, 7):
        host, ok_h = kd._tree_sum(e[:m].copy(), ed.batch_add)
        dev, ok_d = kd._tree_sum(torch.from_numpy(e[:m].copy()).cuda(), ed.batch_add)
        assert ok_h and ok_d and bytes(dev.cpu().numpy()) == bytes(np.asarray(host)), m
    bad = np.asarray(bls.g1_commit(k)[0]).copy()
    bad[2] = 0xFF
    _, ok = kd._tree_sum(torch.from_numpy(bad[:5]).cuda(), lambda a, b: bls.ENGINE.add(1, a, b))
    assert not ok
