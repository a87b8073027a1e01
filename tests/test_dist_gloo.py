"""World-size-2 gloo tests (CPU) of the multi-GPU driver: sharding arithmetic and the MSM exchange
step (all-gather of encoded partial points + local combine).  The single-device MSM is injected --
here the oracle stands in for the HIP engine, which has no GPU to run on in this container."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kyber_amd import dist as kd


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [kd.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _oracle_ed_msm(scalars, points):
    from tests import _oracle_c as OC

    s = np.asarray(scalars, dtype=np.uint8).reshape(-1, 32)
    p = np.asarray(points, dtype=np.uint8).reshape(-1, 32)
    out, rc = OC.ed_msm(s, p)
    st = np.zeros(len(s), dtype=np.uint8)
    if rc:
        st[rc - 1] = 1
        out = np.zeros(32, dtype=np.uint8)
    return out, st


def _oracle_bls_g1_msm(scalars, points):
    from oracle import bls12381 as O

    s = np.asarray(scalars, dtype=np.uint8).reshape(-1, 32)
    p = np.asarray(points, dtype=np.uint8).reshape(-1, 48)
    st = np.zeros(len(s), dtype=np.uint8)
    try:
        out = O.g1_msm_bytes([bytes(x) for x in s], [bytes(x) for x in p])
    except O.DecodeError:
        st[:] = 1
        out = bytes(48)
    return np.frombuffer(out, dtype=np.uint8).copy(), st


def _oracle_bls_g1_add(a, b):
    """Batched Point.Add stand-in (what Engine.add does on the device) for the tree combine."""
    from oracle import bls12381 as O

    a = np.asarray(a, dtype=np.uint8).reshape(-1, 48)
    b = np.asarray(b, dtype=np.uint8).reshape(-1, 48)
    out = np.stack([np.frombuffer(O.g1_compress(O.g1_add(O.g1_decompress(bytes(x)), O.g1_decompress(bytes(y)))),
                                  dtype=np.uint8) for x, y in zip(a, b)])
    return out, np.zeros(len(a), dtype=np.uint8)


def test_tree_sum_odd_and_even_counts():
    from oracle import bls12381 as O

    pts = [O.g1_mul(k, O.G1_GEN) for k in (3, 5, 7, 11, 13, 17, 19)]
    enc = np.stack([np.frombuffer(O.g1_compress(p), dtype=np.uint8) for p in pts])
    for m in (1, 2, 3, 4, 7):
        out, ok = kd._tree_sum(enc[:m], _oracle_bls_g1_add)
        assert ok and bytes(out) == O.g1_compress(O.g1_mul(sum((3, 5, 7, 11, 13, 17, 19)[:m]), O.G1_GEN))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests import _oracle_c as OC
        from oracle import bls12381 as OB

        # ---- Ed25519: 41 points sharded 21 / 20
        n = 41
        rng = np.random.default_rng(7)
        s = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        s[:, 31] &= 0x0F
        h = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        h[:, 31] &= 0x0F
        P = OC.ed_mul_base(h, threads=1)
        lo, hi = kd.shard_range(n, rank, world)
        out, ok = kd.msm_allgather(s[lo:hi], P[lo:hi], _oracle_ed_msm, 32, True)
        full, rc = OC.ed_msm(s, P)
        res = {"ed_ok": bool(ok) and rc == 0 and bytes(out) == bytes(full)}
        # ---- a bad point on rank 1 only must zero the result on every rank
        P2 = P.copy()
        P2[n - 1] = np.frombuffer(bytes([2]) + bytes(31), dtype=np.uint8)
        out, ok = kd.msm_allgather(s[lo:hi], P2[lo:hi], _oracle_ed_msm, 32, True)
        res["ed_bad"] = (not ok) and not np.asarray(out).any()
        # ---- BLS12-381 G1 (big-endian scalars, 48-byte points): 6 points sharded 3 / 3
        ks = [int.from_bytes(bytes(s[i]), "big") % OB.R for i in range(6)]
        pts = [OB.g1_compress(OB.g1_mul(int.from_bytes(bytes(h[i]), "big") % OB.R, OB.G1_GEN)) for i in range(6)]
        kb = np.stack([np.frombuffer(k.to_bytes(32, "big"), dtype=np.uint8) for k in ks])
        pb = np.stack([np.frombuffer(p, dtype=np.uint8) for p in pts])
        lo, hi = kd.shard_range(6, rank, world)
        out, ok = kd.msm_allgather(kb[lo:hi], pb[lo:hi], _oracle_bls_g1_msm, 48, False)
        exp = OB.g1_msm_bytes([k.to_bytes(32, "big") for k in ks], pts)
        res["bls_ok"] = bool(ok) and bytes(out) == exp
        # the same exchange with the partial points combined by the batched-add tree (what the suite wrappers use)
        out, ok = kd.msm_allgather(kb[lo:hi], pb[lo:hi], _oracle_bls_g1_msm, 48, False, combine_add=_oracle_bls_g1_add)
        res["bls_tree_ok"] = bool(ok) and bytes(out) == exp
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_msm_allgather_world2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res in results:
        assert all(res.values()), (rank, res)


def _worker_wide(rank, world, port, q):
    """world 4 / 8: uneven shards (n not a multiple of world, some ranks one point longer), a rejected point on a MIDDLE
    rank, an empty shard (n < world), and the partial points combined by the add tree with an odd count at some level."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests import _oracle_c as OC

        res = {}
        for n in (world * 5 + 3, world + 1, world - 1):
            rng = np.random.default_rng(100 + n)
            s = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
            s[:, 31] &= 0x0F
            h = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
            h[:, 31] &= 0x0F
            P = OC.ed_mul_base(h, threads=1)
            lo, hi = kd.shard_range(n, rank, world)
            full, rc = OC.ed_msm(s, P)
            out, ok = kd.msm_allgather(s[lo:hi], P[lo:hi], _oracle_ed_msm, 32, True)
            res["ed_n%d" % n] = bool(ok) and rc == 0 and bytes(out) == bytes(full)
            out, ok = kd.msm_allgather(s[lo:hi], P[lo:hi], _oracle_ed_msm, 32, True, combine_add=_oracle_ed_add)
            res["ed_tree_n%d" % n] = bool(ok) and bytes(out) == bytes(full)
        # a bad point inside the shard of a middle rank: every rank must report failure and hold zero bytes
        n = world * 5 + 3
        rng = np.random.default_rng(100 + n)
        s = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        s[:, 31] &= 0x0F
        h = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        h[:, 31] &= 0x0F
        P = OC.ed_mul_base(h, threads=1)
        mid_lo, mid_hi = kd.shard_range(n, world // 2, world)
        P[mid_lo + 1] = np.frombuffer(bytes([2]) + bytes(31), dtype=np.uint8)
        lo, hi = kd.shard_range(n, rank, world)
        for name, kw in (("msm", {}), ("tree", {"combine_add": _oracle_ed_add})):
            out, ok = kd.msm_allgather(s[lo:hi], P[lo:hi], _oracle_ed_msm, 32, True, **kw)
            res["bad_mid_" + name] = (not ok) and not np.asarray(out).any()
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def _oracle_ed_add(a, b):
    from oracle import ed25519 as O

    a = np.asarray(a, dtype=np.uint8).reshape(-1, 32)
    b = np.asarray(b, dtype=np.uint8).reshape(-1, 32)
    out = np.stack([np.frombuffer(O.encode(O.add(O.decode(bytes(x)), O.decode(bytes(y)))), dtype=np.uint8) for x, y in zip(a, b)])
    return out, np.zeros(len(a), dtype=np.uint8)


@pytest.mark.parametrize("world", [4, 8])
def test_msm_allgather_world4_and_8_gloo(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_wide, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _ in results) == list(range(world))
    for rank, res in results:
        assert all(res.values()), (rank, res)
