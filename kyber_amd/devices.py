"""Several GPUs of one node behind the host-buffer calls (include/kyber_hip.h: kyb_init_devices / kyb_set_devices).

One process, one host thread + device context per slice of a batch; the reference's callers (share/poly.go:143-149,
sign/bdn/bdn.go:126-161) are single-process loops, so this is the form a cgo suite uses.  kyber_amd/dist.py is the
one-process-per-GPU variant (torch.distributed, RCCL all-gather of the partial points of an MSM).
"""
from __future__ import annotations

import ctypes

from ._lib import check, load


def init_devices(ndev: int) -> None:
    """Use devices 0 .. ndev-1 for every host-buffer batch call (contexts are created now)."""
    check(load().kyb_init_devices(ndev), "kyb_init_devices")


def set_devices(devices) -> None:
    """Use exactly these HIP devices (an entry may repeat: its slices then run one after the other);
    an empty list restores single-device behaviour."""
    arr = (ctypes.c_int * max(1, len(devices)))(*devices)
    check(load().kyb_set_devices(ctypes.cast(arr, ctypes.c_void_p), len(devices)), "kyb_set_devices")


def get_devices() -> list:
    arr = (ctypes.c_int * 64)()
    n = load().kyb_get_devices(ctypes.cast(arr, ctypes.c_void_p), 64)
    return list(arr[:n])


def set_shard_threshold(min_units: int) -> None:
    """Host batches below this many units stay on the caller's device (default 16384)."""
    check(load().kyb_set_shard_threshold(min_units), "kyb_set_shard_threshold")


def shard_range(n: int, rank: int, world: int):
    """[lo, hi) of shard `rank` of `world` over n units: the library's own rule (= kyber_amd.dist.shard_range)."""
    lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
    load().kyb_shard_range(n, rank, world, ctypes.byref(lo), ctypes.byref(hi))
    return lo.value, hi.value
