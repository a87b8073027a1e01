"""Host-buffer entry points whose transfers bounce through the page-locked slots (context.h StageBuf::bounce: >= 4 MB,
8 MB pieces, double-buffered) against the same calls on device-resident tensors: sizes that are not multiples of a
piece, more than two pieces, and results larger than inputs."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rows(label, n, w):
    return np.frombuffer(hashlib.shake_256(label).digest(n * w), dtype=np.uint8).reshape(n, w).copy()


def test_pairing_suite_host_paths_match_resident_paths():
    import torch

    from kyber_amd.pairing import bls12381 as bls, bn256 as bn

    for m, n in ((bls, 43_210), (bn, 70_001)):      # GT output 24.9 MB / 26.9 MB: four pieces, the last one ragged
        k = _rows(b"stage/k/" + m.__name__.encode(), n, 32)
        k[:, 0] &= 0x3F
        P, st1 = m.g1_commit(k)                      # host in (1.4 / 2.2 MB: direct copy), host out
        Q, st2 = m.g2_commit(k[::-1].copy())
        assert not np.asarray(st1).any() and not np.asarray(st2).any()
        dk, dP, dQ = (torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in (k, P, Q))
        gt_h, st_h = m.batch_pair(P, Q)              # host buffers
        gt_d, st_d = m.batch_pair(dP, dQ)            # resident
        assert not np.asarray(st_h).any() and not st_d.any().item()
        assert (np.asarray(gt_h) == gt_d.cpu().numpy()).all()
        r_h, _ = m.g2_batch_mul(k, Q)                # 5.5 / 9 MB in and out: bounce both ways
        r_d, _ = m.g2_batch_mul(dk, dQ)
        assert (np.asarray(r_h) == r_d.cpu().numpy()).all()
        # corrupt one element in the middle of a later piece: its status and zeroed output arrive at the right row
        bad = np.array(Q, copy=True)
        bad[n - 7] = 0
        bad[n - 7, 31] = 5
        r_b, st_b = m.g2_batch_mul(k, bad)
        assert np.flatnonzero(np.asarray(st_b)).tolist() == [n - 7] and not np.asarray(r_b)[n - 7].any()
        assert (np.delete(np.asarray(r_b), n - 7, 0) == np.delete(np.asarray(r_h), n - 7, 0)).all()


def test_msm_host_path_matches_resident_path():
    import torch

    from kyber_amd.pairing import bls12381 as bls

    n = 150_001                                       # 14.4 MB of compressed points + scalars
    k = _rows(b"stage/msm/k", n, 32)
    k[:, 0] &= 0x3F
    P, _ = bls.g1_commit(_rows(b"stage/msm/h", n, 32) & 0x3F)
    out_h, st = bls.g1_msm(k, P, bls.F_TRUSTED(0))
    out_d, _ = bls.g1_msm(torch.from_numpy(k).cuda(), torch.from_numpy(np.ascontiguousarray(P)).cuda(), bls.F_TRUSTED(0))
    assert not np.asarray(st).any() and bytes(np.asarray(out_h)) == bytes(out_d.cpu().numpy())
