"""Multi-GPU driver: one process per GPU (``torch.distributed``; backend "nccl" is RCCL over xGMI on
ROCm, "gloo" for the CPU tests).

What shards and how:
  * batched scalar-mul / pairing / pairing-check: independent units -> every rank processes its
    own contiguous slice, NO data-path collective (``shard_range`` + the batch API);
  * node-wide MSM  sum_i k_i P_i : shard the POINTS.  Each rank runs the full Pippenger pipeline on
    its slice and obtains one encoded partial point (32 / 48 / 96 / 64 / 128 bytes).  Elliptic
    curve addition is not an RCCL reduce op, so the exchange step is ONE all-gather of the
    encoded partials (world x <=128 bytes -- latency bound, never bucket arrays), after which
    every rank adds the ``world`` partial points locally (an MSM with unit scalars) and holds
    the identical result.

The local compute is injected (``local_msm``) so the exchange logic can be covered by the
world_size-2 gloo tests in this GPU-less container with the oracle standing in; the defaults are
the HIP engine, which fails loudly when the library is missing.
"""
from __future__ import annotations

from typing import Callable

import numpy as np


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous slice [lo, hi) of n units owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _unit_scalars(world: int, little_endian: bool) -> np.ndarray:
    s = np.zeros((world, 32), dtype=np.uint8)
    s[:, 0 if little_endian else 31] = 1
    return s


def _tree_sum(points, add_fn: Callable, bad=None):
    """Sum the rows of `points` (encoded group elements) with log2(n) batched Point.Add calls.  Returns (enc, ok).
    CUDA tensors stay on the device: the adds are enqueued on the current stream and the "any input rejected" bit is
    read once, at the end (one synchronisation for the whole combine instead of one per level and a host round trip
    of the partial points)."""
    if type(points).__module__.startswith("torch"):
        import torch

        pts = points.contiguous()
        bad = torch.zeros((), dtype=torch.bool, device=pts.device) if bad is None else bad  # (a rank already saw a bad input)
        while pts.shape[0] > 1:
            m = pts.shape[0] // 2
            out, st = add_fn(pts[0:2 * m:2].contiguous(), pts[1:2 * m:2].contiguous())
            bad = bad | st.any()
            out = out.view(m, -1)
            pts = torch.cat([out, pts[2 * m:]], dim=0) if pts.shape[0] % 2 else out
        return pts[0], not bool(bad.item())
    pts = np.ascontiguousarray(points)
    ok = True
    while pts.shape[0] > 1:
        m = pts.shape[0] // 2
        out, st = add_fn(pts[0:2 * m:2].copy(), pts[1:2 * m:2].copy())
        ok = ok and not bool(np.asarray(st).any())
        out = np.asarray(out).reshape(m, -1)
        pts = np.concatenate([out, pts[2 * m:]], axis=0) if pts.shape[0] % 2 else out
    return pts[0], ok


def msm_allgather(scalars_shard, points_shard, local_msm: Callable, point_len: int, little_endian: bool,
                  group=None, combine_msm: Callable | None = None, combine_add: Callable | None = None):
    """Node-wide MSM over the shards held by the ranks of `group`.

    scalars_shard / points_shard: this rank's slice (host numpy/bytes, or CUDA uint8 tensors when
    the backend is nccl).  local_msm(scalars, points) -> (encoded_point, status) is the single-GPU
    MSM (e.g. ``edwards25519.msm`` or ``bls12381.g1_msm``).  Returns (encoded_point, ok) where
    ok is False iff any rank rejected an input (then the point is all-zero bytes), identical on
    every rank.  The gathered partial points (one per rank) are summed by combine_add -- a batched Point.Add used as a
    log2(world) tree, ~0.1 ms -- when given, else by combine_msm (default: local_msm) with unit scalars (a full
    pipeline launch: ~2 ms of fixed latency for a handful of points)."""
    import torch
    import torch.distributed as dist

    if combine_msm is None:
        combine_msm = local_msm

    world = dist.get_world_size(group)
    part, st = local_msm(scalars_shard, points_shard)
    is_t = type(part).__module__.startswith("torch")
    dev = part.device if is_t else torch.device("cpu")
    on_gpu = is_t and dev.type == "cuda"
    # payload: encoded partial point + one "bad input seen" byte.  On the GPU everything up to the final verdict is
    # enqueued -- the status reduction, the all-gather (RCCL on the same stream), the combine -- and the host reads ONE
    # boolean at the end.
    buf = torch.zeros(point_len + 1, dtype=torch.uint8, device=dev)
    buf[:point_len] = part.view(-1) if is_t else torch.from_numpy(np.frombuffer(bytes(part), dtype=np.uint8).copy())
    if on_gpu:
        buf[point_len] = st.any().to(torch.uint8)
    else:
        buf[point_len] = 1 if (bool(st.any().item()) if is_t else bool(np.asarray(st).any())) else 0
    gathered = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf, group=group)
    allp = torch.stack(gathered)
    pts = allp[:, :point_len].contiguous()
    if on_gpu and combine_add is not None:
        enc, ok = _tree_sum(pts, combine_add, bad=allp[:, point_len].any())
        return (enc if ok else torch.zeros(point_len, dtype=torch.uint8, device=dev)), ok
    if bool(allp[:, point_len].any().item()):
        zero = torch.zeros(point_len, dtype=torch.uint8, device=dev)
        return (zero if is_t else zero.numpy()), False
    if combine_add is not None:
        enc, ok = _tree_sum(pts.cpu().numpy(), combine_add)
        return (torch.from_numpy(np.ascontiguousarray(enc)).to(dev) if is_t else enc), ok
    ones = _unit_scalars(world, little_endian)
    if is_t:
        out, st2 = combine_msm(torch.from_numpy(ones).to(dev), pts)
        ok = not bool(st2.any().item())
    else:
        out, st2 = combine_msm(ones, pts.numpy())
        ok = not bool(np.asarray(st2).any())
    return out, ok


def ed25519_msm(scalars_shard, points_shard, group=None):
    from .group import edwards25519 as ed

    return msm_allgather(scalars_shard, points_shard, ed.msm, 32, True, group, combine_add=ed.batch_add)


def bls12381_g1_msm(scalars_shard, points_shard, group=None, flags: int = 0):
    """flags: the shard's input flags (F_TRUSTED(0) / F_UNCOMPRESSED, kyber_amd.pairing._engine)."""
    from .pairing import bls12381 as m

    return msm_allgather(scalars_shard, points_shard, lambda s, p: m.g1_msm(s, p, flags), m.G1_LEN, False, group,
                         combine_add=lambda a, b: m.ENGINE.add(1, a, b))


def bls12381_g2_msm(scalars_shard, points_shard, group=None, flags: int = 0):
    """flags: the shard's input flags (F_TRUSTED(0) / F_UNCOMPRESSED, kyber_amd.pairing._engine)."""
    from .pairing import bls12381 as m

    return msm_allgather(scalars_shard, points_shard, lambda s, p: m.g2_msm(s, p, flags), m.G2_LEN, False, group,
                         combine_add=lambda a, b: m.ENGINE.add(2, a, b))


def bn256_g1_msm(scalars_shard, points_shard, group=None, flags: int = 0):
    """flags: the shard's input flags (F_TRUSTED(0) / F_UNCOMPRESSED, kyber_amd.pairing._engine)."""
    from .pairing import bn256 as m

    return msm_allgather(scalars_shard, points_shard, lambda s, p: m.g1_msm(s, p, flags), m.G1_LEN, False, group,
                         combine_add=lambda a, b: m.ENGINE.add(1, a, b))


def bn256_g2_msm(scalars_shard, points_shard, group=None, flags: int = 0):
    """flags: the shard's input flags (F_TRUSTED(0) / F_UNCOMPRESSED, kyber_amd.pairing._engine)."""
    from .pairing import bn256 as m

    return msm_allgather(scalars_shard, points_shard, lambda s, p: m.g2_msm(s, p, flags), m.G2_LEN, False, group,
                         combine_add=lambda a, b: m.ENGINE.add(2, a, b))


def bn254_g1_msm(scalars_shard, points_shard, group=None, flags: int = 0):
    """flags: the shard's input flags (F_TRUSTED(0), kyber_amd.pairing._engine)."""
    from .pairing import bn254 as m

    return msm_allgather(scalars_shard, points_shard, lambda s, p: m.g1_msm(s, p, flags), m.G1_LEN, False, group,
                         combine_add=lambda a, b: m.ENGINE.add(1, a, b))


def bn254_g2_msm(scalars_shard, points_shard, group=None, flags: int = 0):
    """flags: the shard's input flags (F_TRUSTED(0): the points were unmarshalled, i.e. subgroup-checked, before)."""
    from .pairing import bn254 as m

    return msm_allgather(scalars_shard, points_shard, lambda s, p: m.g2_msm(s, p, flags), m.G2_LEN, False, group,
                         combine_add=lambda a, b: m.ENGINE.add(2, a, b))
