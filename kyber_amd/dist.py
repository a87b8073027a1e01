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


def msm_allgather(scalars_shard, points_shard, local_msm: Callable, point_len: int, little_endian: bool,
                  group=None, combine_msm: Callable | None = None):
    """Node-wide MSM over the shards held by the ranks of `group`.

    scalars_shard / points_shard: this rank's slice (host numpy/bytes, or CUDA uint8 tensors when
    the backend is nccl).  local_msm(scalars, points) -> (encoded_point, status) is the single-GPU
    MSM (e.g. ``edwards25519.msm`` or ``bls12381.g1_msm``).  Returns (encoded_point, ok) where
    ok is False iff any rank rejected an input (then the point is all-zero bytes), identical on
    every rank.  combine_msm (default: local_msm) sums the gathered partial points; the suite wrappers below pass
    one that marks its inputs as trusted, since the partials are outputs of this library."""
    import torch
    import torch.distributed as dist

    if combine_msm is None:
        combine_msm = local_msm

    world = dist.get_world_size(group)
    part, st = local_msm(scalars_shard, points_shard)
    is_t = type(part).__module__.startswith("torch")
    dev = part.device if is_t else torch.device("cpu")
    bad_local = bool(st.any().item()) if is_t else bool(np.asarray(st).any())
    # payload: encoded partial point + one "bad input seen" byte
    buf = torch.zeros(point_len + 1, dtype=torch.uint8, device=dev)
    buf[:point_len] = part.view(-1) if is_t else torch.from_numpy(np.frombuffer(bytes(part), dtype=np.uint8).copy())
    buf[point_len] = 1 if bad_local else 0
    gathered = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf, group=group)
    allp = torch.stack(gathered)
    if bool(allp[:, point_len].any().item()):
        zero = torch.zeros(point_len, dtype=torch.uint8, device=dev)
        return (zero if is_t else zero.numpy()), False
    pts = allp[:, :point_len].contiguous()
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

    return msm_allgather(scalars_shard, points_shard, ed.msm, 32, True, group)


def bls12381_g1_msm(scalars_shard, points_shard, group=None, flags: int = 0):
    """flags: the shard's input flags (F_TRUSTED(0) / F_UNCOMPRESSED, kyber_amd.pairing._engine)."""
    from .pairing import bls12381 as m
    from .pairing._engine import F_TRUSTED

    return msm_allgather(scalars_shard, points_shard, lambda s, p: m.g1_msm(s, p, flags), m.G1_LEN, False, group,
                         combine_msm=lambda s, p: m.g1_msm(s, p, F_TRUSTED(0)))


def bls12381_g2_msm(scalars_shard, points_shard, group=None, flags: int = 0):
    """flags: the shard's input flags (F_TRUSTED(0) / F_UNCOMPRESSED, kyber_amd.pairing._engine)."""
    from .pairing import bls12381 as m
    from .pairing._engine import F_TRUSTED

    return msm_allgather(scalars_shard, points_shard, lambda s, p: m.g2_msm(s, p, flags), m.G2_LEN, False, group,
                         combine_msm=lambda s, p: m.g2_msm(s, p, F_TRUSTED(0)))


def bn256_g1_msm(scalars_shard, points_shard, group=None, flags: int = 0):
    """flags: the shard's input flags (F_TRUSTED(0) / F_UNCOMPRESSED, kyber_amd.pairing._engine)."""
    from .pairing import bn256 as m
    from .pairing._engine import F_TRUSTED

    return msm_allgather(scalars_shard, points_shard, lambda s, p: m.g1_msm(s, p, flags), m.G1_LEN, False, group,
                         combine_msm=lambda s, p: m.g1_msm(s, p, F_TRUSTED(0)))


def bn256_g2_msm(scalars_shard, points_shard, group=None, flags: int = 0):
    """flags: the shard's input flags (F_TRUSTED(0) / F_UNCOMPRESSED, kyber_amd.pairing._engine)."""
    from .pairing import bn256 as m
    from .pairing._engine import F_TRUSTED

    return msm_allgather(scalars_shard, points_shard, lambda s, p: m.g2_msm(s, p, flags), m.G2_LEN, False, group,
                         combine_msm=lambda s, p: m.g2_msm(s, p, F_TRUSTED(0)))
