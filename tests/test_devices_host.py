"""Multi-device layer of the C ABI, the parts that need no GPU: the partition rule and the argument checks."""
import random

import pytest

from kyber_amd import devices, dist
from kyber_amd._lib import KyberHipError


def test_shard_range_is_the_rule_the_multi_process_path_uses():
    rng = random.Random(7)
    cases = [(0, 1), (0, 8), (1, 8), (7, 8), (8, 8), (9, 8), (1 << 20, 8), ((1 << 20) + 5, 7), (65536, 3)]
    cases += [(rng.randrange(0, 1 << 22), rng.randrange(1, 17)) for _ in range(200)]
    for n, world in cases:
        prev = 0
        for r in range(world):
            lo, hi = devices.shard_range(n, r, world)
            assert (lo, hi) == dist.shard_range(n, r, world), (n, r, world)
            assert lo == prev and hi - lo in (n // world, n // world + 1)
            prev = hi
        assert prev == n
    assert devices.shard_range(10, 5, 3) == (0, 0)  # out-of-range rank: empty


def test_device_set_argument_checks_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("covered by the GPU test")
    with pytest.raises(KyberHipError):
        devices.set_devices([0])  # no device is visible here
    with pytest.raises(KyberHipError):
        devices.init_devices(0)
    devices.set_devices([])       # clearing the set is always allowed
    assert devices.get_devices() == []
