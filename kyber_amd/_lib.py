"""ctypes binding of libkyberhip.so (the C ABI in include/kyber_hip.h).

Fails loudly when the library is absent or a symbol is missing: there is no
CPU fallback in the product path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KYBER_HIP_LIB") or os.path.join(_HERE, "lib", "libkyberhip.so")  # override: A/B builds

KYB_F_VARTIME = 1
KYB_F_UNIFORM = 8  # Ed25519: scalar-independent addresses and control flow (include/kyber_hip.h)
ST_OK, ST_BAD_POINT, ST_NOT_IN_SUBGROUP = 0, 1, 2


class KyberHipError(RuntimeError):
    pass


_vp, _sz, _u32, _int = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int

# name -> argtypes; restype is int unless listed in _RESTYPES
SIGNATURES = {
    "kyb_version": [],
    "kyb_last_error": [],
    "kyb_device_count": [],
    "kyb_init": [],
    "kyb_shutdown": [],
    "kyb_stream_release": [_vp],
    "kyb_init_devices": [_int],
    "kyb_set_devices": [_vp, _int],
    "kyb_get_devices": [_vp, _int],
    "kyb_set_shard_threshold": [_sz],
    "kyb_shard_range": [_sz, _int, _int, _vp, _vp],
    "kyb_ed25519_mul_base": [_sz, _vp, _vp, _u32],
    "kyb_ed25519_mul_base_dev": [_sz, _vp, _vp, _u32, _vp],
    "kyb_ed25519_mul": [_sz, _vp, _vp, _vp, _vp, _u32],
    "kyb_ed25519_mul_dev": [_sz, _vp, _vp, _vp, _vp, _u32, _vp],
    "kyb_ed25519_mul_same_base": [_sz, _vp, _vp, _vp, _vp, _u32],
    "kyb_ed25519_debug_base_table": [_vp],
    "kyb_bls12381_debug_vkey_stats": [_vp, _vp],
    "kyb_ed25519_add": [_sz, _vp, _vp, _vp, _vp],
    "kyb_ed25519_add_dev": [_sz, _vp, _vp, _vp, _vp, _vp],
    "kyb_ed25519_hash": [_sz, _vp, _sz, _vp, _sz, _vp],
    "kyb_ed25519_hash_dev": [_sz, _vp, _sz, _vp, _sz, _vp, _vp],
    "kyb_ed25519_msm": [_sz, _vp, _vp, _vp, _vp],
    "kyb_ed25519_msm_flags": [_sz, _vp, _vp, _vp, _vp, _u32],
    "kyb_ed25519_msm_dev": [_sz, _vp, _vp, _vp, _vp, _vp],
    "kyb_bls12381_g1_msm": [_sz, _vp, _vp, _vp, _vp, _u32],
    "kyb_bls12381_g2_msm": [_sz, _vp, _vp, _vp, _vp, _u32],
    "kyb_bls12381_g1_msm_dev": [_sz, _vp, _vp, _vp, _vp, _u32, _vp],
    "kyb_bls12381_g2_msm_dev": [_sz, _vp, _vp, _vp, _vp, _u32, _vp],
    "kyb_bn256_g1_msm": [_sz, _vp, _vp, _vp, _vp, _u32],
    "kyb_bn256_g2_msm": [_sz, _vp, _vp, _vp, _vp, _u32],
    "kyb_bn256_g1_msm_dev": [_sz, _vp, _vp, _vp, _vp, _u32, _vp],
    "kyb_bn256_g2_msm_dev": [_sz, _vp, _vp, _vp, _vp, _u32, _vp],
    "kyb_bls12381_g1_mul": [_sz, _vp, _vp, _vp, _vp, _u32],
    "kyb_bls12381_g2_mul": [_sz, _vp, _vp, _vp, _vp, _u32],
    "kyb_bls12381_g1_mul_same_base": [_sz, _vp, _vp, _vp, _vp, _u32],
    "kyb_bls12381_g2_mul_same_base": [_sz, _vp, _vp, _vp, _vp, _u32],
    "kyb_bls12381_g1_mul_dev": [_sz, _vp, _vp, _sz, _vp, _vp, _u32, _vp],
    "kyb_bls12381_g2_mul_dev": [_sz, _vp, _vp, _sz, _vp, _vp, _u32, _vp],
    "kyb_bls12381_g1_add": [_sz, _vp, _vp, _vp, _vp],
    "kyb_bls12381_g2_add": [_sz, _vp, _vp, _vp, _vp],
    "kyb_bls12381_g1_add_dev": [_sz, _vp, _vp, _vp, _vp, _vp],
    "kyb_bls12381_g2_add_dev": [_sz, _vp, _vp, _vp, _vp, _vp],
    "kyb_bls12381_pair": [_sz, _vp, _vp, _vp, _vp, _u32],
    "kyb_bls12381_hash_g1": [_sz, _vp, _sz, _vp, _sz, _vp, _vp],
    "kyb_bls12381_hash_g2": [_sz, _vp, _sz, _vp, _sz, _vp, _vp],
    "kyb_bls12381_hash_g1_dev": [_sz, _vp, _sz, _vp, _sz, _vp, _vp, _vp],
    "kyb_bls12381_hash_g2_dev": [_sz, _vp, _sz, _vp, _sz, _vp, _vp, _vp],
    "kyb_bls12381_verify_g1": [_sz, _vp, _vp, _sz, _vp, _sz, _vp, _vp, _vp, _u32],
    "kyb_bls12381_verify_g1_dev": [_sz, _vp, _vp, _sz, _vp, _sz, _vp, _vp, _vp, _u32, _vp],
    "kyb_bls12381_verify_g1_same_key": [_sz, _vp, _vp, _sz, _vp, _sz, _vp, _vp, _vp, _u32],
    "kyb_bls12381_verify_g1_same_key_dev": [_sz, _vp, _vp, _sz, _vp, _sz, _vp, _vp, _vp, _u32, _vp],
    "kyb_bls12381_verify_g1_same_msg": [_sz, _vp, _vp, _sz, _vp, _sz, _vp, _vp, _vp, _u32],
    "kyb_bls12381_verify_g1_same_msg_dev": [_sz, _vp, _vp, _sz, _vp, _sz, _vp, _vp, _vp, _u32, _vp],
    "kyb_bls12381_verify_g2": [_sz, _vp, _vp, _sz, _vp, _sz, _vp, _vp, _vp, _u32],
    "kyb_bls12381_verify_g2_dev": [_sz, _vp, _vp, _sz, _vp, _sz, _vp, _vp, _vp, _u32, _vp],
    "kyb_bls12381_gt_mul": [_sz, _vp, _vp, _vp, _vp],
    "kyb_bls12381_gt_mul_dev": [_sz, _vp, _vp, _vp, _vp, _vp],
    "kyb_bls12381_pair_dev": [_sz, _vp, _vp, _vp, _vp, _u32, _vp],
    "kyb_bls12381_pair_check": [_sz, _vp, _vp, _vp, _vp, _vp, _vp, _u32],
    "kyb_bls12381_pair_check_dev": [_sz, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp],
    "kyb_bn256_g1_mul": [_sz, _vp, _vp, _vp, _vp, _u32],
    "kyb_bn256_g2_mul": [_sz, _vp, _vp, _vp, _vp, _u32],
    "kyb_bn256_g1_mul_same_base": [_sz, _vp, _vp, _vp, _vp, _u32],
    "kyb_bn256_g2_mul_same_base": [_sz, _vp, _vp, _vp, _vp, _u32],
    "kyb_bn256_g1_mul_dev": [_sz, _vp, _vp, _sz, _vp, _vp, _u32, _vp],
    "kyb_bn256_g2_mul_dev": [_sz, _vp, _vp, _sz, _vp, _vp, _u32, _vp],
    "kyb_bn256_g1_add": [_sz, _vp, _vp, _vp, _vp],
    "kyb_bn256_g2_add": [_sz, _vp, _vp, _vp, _vp],
    "kyb_bn256_g1_add_dev": [_sz, _vp, _vp, _vp, _vp, _vp],
    "kyb_bn256_g2_add_dev": [_sz, _vp, _vp, _vp, _vp, _vp],
    "kyb_bn256_pair": [_sz, _vp, _vp, _vp, _vp, _u32],
    "kyb_bn256_hash_g1": [_sz, _vp, _sz, _vp, _vp],
    "kyb_bn256_hash_g1_dev": [_sz, _vp, _sz, _vp, _vp, _vp],
    "kyb_bn256_hash_g1_svdw": [_sz, _vp, _sz, _vp, _sz, _vp, _vp],
    "kyb_bn256_hash_g1_svdw_dev": [_sz, _vp, _sz, _vp, _sz, _vp, _vp, _vp],
    "kyb_bn256_gt_mul": [_sz, _vp, _vp, _vp, _vp],
    "kyb_bn256_gt_mul_dev": [_sz, _vp, _vp, _vp, _vp, _vp],
    "kyb_bn256_pair_dev": [_sz, _vp, _vp, _vp, _vp, _u32, _vp],
    "kyb_bn256_pair_check": [_sz, _vp, _vp, _vp, _vp, _vp, _vp, _u32],
    "kyb_bn256_pair_check_dev": [_sz, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp],
    "kyb_ed25519_unmarshal": [_sz, _vp, _vp, _vp],
    "kyb_ed25519_unmarshal_dev": [_sz, _vp, _vp, _vp, _vp],
    "kyb_bls12381_g1_unmarshal": [_sz, _vp, _vp, _vp, _u32],
    "kyb_bls12381_g1_unmarshal_dev": [_sz, _vp, _vp, _vp, _u32, _vp],
    "kyb_bls12381_g2_unmarshal": [_sz, _vp, _vp, _vp, _u32],
    "kyb_bls12381_g2_unmarshal_dev": [_sz, _vp, _vp, _vp, _u32, _vp],
    "kyb_bn256_g1_unmarshal": [_sz, _vp, _vp, _vp, _u32],
    "kyb_bn256_g1_unmarshal_dev": [_sz, _vp, _vp, _vp, _u32, _vp],
    "kyb_bn256_g2_unmarshal": [_sz, _vp, _vp, _vp, _u32],
    "kyb_bn256_g2_unmarshal_dev": [_sz, _vp, _vp, _vp, _u32, _vp],
    "kyb_ed25519_poly_eval": [_sz, _vp, _sz, _vp, _vp, _vp],
    "kyb_ed25519_poly_eval_dev": [_sz, _vp, _sz, _vp, _vp, _vp, _vp],
    "kyb_bls12381_g1_poly_eval": [_sz, _vp, _sz, _vp, _vp, _vp, _u32],
    "kyb_bls12381_g2_poly_eval": [_sz, _vp, _sz, _vp, _vp, _vp, _u32],
    "kyb_bls12381_g1_poly_eval_dev": [_sz, _vp, _sz, _vp, _vp, _vp, _u32, _vp],
    "kyb_bls12381_g2_poly_eval_dev": [_sz, _vp, _sz, _vp, _vp, _vp, _u32, _vp],
    "kyb_bn256_g1_poly_eval": [_sz, _vp, _sz, _vp, _vp, _vp, _u32],
    "kyb_bn256_g2_poly_eval": [_sz, _vp, _sz, _vp, _vp, _vp, _u32],
    "kyb_bn256_g1_poly_eval_dev": [_sz, _vp, _sz, _vp, _vp, _vp, _u32, _vp],
    "kyb_bn256_g2_poly_eval_dev": [_sz, _vp, _sz, _vp, _vp, _vp, _u32, _vp],
    "kyb_bn254_g1_msm": [_sz, _vp, _vp, _vp, _vp, _u32],
    "kyb_bn254_g2_msm": [_sz, _vp, _vp, _vp, _vp, _u32],
    "kyb_bn254_g1_msm_dev": [_sz, _vp, _vp, _vp, _vp, _u32, _vp],
    "kyb_bn254_g2_msm_dev": [_sz, _vp, _vp, _vp, _vp, _u32, _vp],
    "kyb_bn254_g1_mul": [_sz, _vp, _vp, _vp, _vp, _u32],
    "kyb_bn254_g2_mul": [_sz, _vp, _vp, _vp, _vp, _u32],
    "kyb_bn254_g1_mul_same_base": [_sz, _vp, _vp, _vp, _vp, _u32],
    "kyb_bn254_g2_mul_same_base": [_sz, _vp, _vp, _vp, _vp, _u32],
    "kyb_bn254_g1_mul_dev": [_sz, _vp, _vp, _sz, _vp, _vp, _u32, _vp],
    "kyb_bn254_g2_mul_dev": [_sz, _vp, _vp, _sz, _vp, _vp, _u32, _vp],
    "kyb_bn254_g1_add": [_sz, _vp, _vp, _vp, _vp],
    "kyb_bn254_g2_add": [_sz, _vp, _vp, _vp, _vp],
    "kyb_bn254_g1_add_dev": [_sz, _vp, _vp, _vp, _vp, _vp],
    "kyb_bn254_g2_add_dev": [_sz, _vp, _vp, _vp, _vp, _vp],
    "kyb_bn254_pair": [_sz, _vp, _vp, _vp, _vp, _u32],
    "kyb_bn254_gt_mul": [_sz, _vp, _vp, _vp, _vp],
    "kyb_bn254_gt_mul_dev": [_sz, _vp, _vp, _vp, _vp, _vp],
    "kyb_bn254_pair_dev": [_sz, _vp, _vp, _vp, _vp, _u32, _vp],
    "kyb_bn254_pair_check": [_sz, _vp, _vp, _vp, _vp, _vp, _vp, _u32],
    "kyb_bn254_pair_check_dev": [_sz, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp],
    "kyb_bn254_g1_unmarshal": [_sz, _vp, _vp, _vp, _u32],
    "kyb_bn254_g1_unmarshal_dev": [_sz, _vp, _vp, _vp, _u32, _vp],
    "kyb_bn254_g2_unmarshal": [_sz, _vp, _vp, _vp, _u32],
    "kyb_bn254_g2_unmarshal_dev": [_sz, _vp, _vp, _vp, _u32, _vp],
    "kyb_ed25519_scalar_poly_eval": [_sz, _vp, _sz, _vp, _vp],
    "kyb_ed25519_scalar_poly_eval_dev": [_sz, _vp, _sz, _vp, _vp, _vp],
    "kyb_bls12381_scalar_poly_eval": [_sz, _vp, _sz, _vp, _vp],
    "kyb_bls12381_scalar_poly_eval_dev": [_sz, _vp, _sz, _vp, _vp, _vp],
    "kyb_bn256_scalar_poly_eval": [_sz, _vp, _sz, _vp, _vp],
    "kyb_bn256_scalar_poly_eval_dev": [_sz, _vp, _sz, _vp, _vp, _vp],
    "kyb_bn254_scalar_poly_eval": [_sz, _vp, _sz, _vp, _vp],
    "kyb_bn254_scalar_poly_eval_dev": [_sz, _vp, _sz, _vp, _vp, _vp],
    "kyb_bn254_g1_poly_eval": [_sz, _vp, _sz, _vp, _vp, _vp, _u32],
    "kyb_bn254_g2_poly_eval": [_sz, _vp, _sz, _vp, _vp, _vp, _u32],
    "kyb_bn254_g1_poly_eval_dev": [_sz, _vp, _sz, _vp, _vp, _vp, _u32, _vp],
    "kyb_bn254_g2_poly_eval_dev": [_sz, _vp, _sz, _vp, _vp, _vp, _u32, _vp],
    "kyb_bn254_hash_g1": [_sz, _vp, _sz, _vp, _sz, _vp, _vp],
    "kyb_bn254_hash_g1_dev": [_sz, _vp, _sz, _vp, _sz, _vp, _vp, _vp],
}
_RESTYPES = {"kyb_last_error": C.c_char_p, "kyb_shard_range": None}

_lib = None


def load() -> C.CDLL:
    """Load (once) and return the library; raise if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KyberHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # PyTorch-ROCm ships its own HIP runtime.  If libkyberhip.so pulled in the system one first, a later
    # `import torch` would find a HIP runtime it did not initialise and report no usable device
    # (torch.cuda.is_available() == False): let torch's copy load first whenever torch is installed.
    try:
        import torch  # noqa: F401
    except ImportError:  # pure host-buffer use without torch
        pass
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise KyberHipError(f"libkyberhip.so lacks symbol {name}") from e
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().kyb_last_error()
        raise KyberHipError(f"{what} failed rc={rc}: {msg.decode() if msg else ''}")
