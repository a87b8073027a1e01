"""Host-side mirror of the reference's ``pairing/bls12381`` suites for the hot path
(pairing.Suite, pairing/pairing.go:8-20; kilic adapter kilic/{g1,g2,gt,suite}.go), backed by the
HIP engine through the C ABI.  No curve or field arithmetic happens in Python.

Wire formats are the adapters' MarshalBinary encodings: scalars 32-byte big-endian (mod.Int),
G1 48-byte / G2 96-byte ZCash compressed, GT 576 bytes.

Batch functions accept host data (bytes / numpy uint8) or device-resident ``torch.uint8`` CUDA
tensors; device inputs are processed on the current stream and results stay on the device.
"""
from ._engine import Engine

ORDER = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001  # kilic/scalar.go:11-12
G1_LEN, G2_LEN, GT_LEN, SCALAR_LEN = 48, 96, 576, 32
G1_BASE = bytes.fromhex(
    "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
G2_BASE = bytes.fromhex(
    "93e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
    "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8")
G1_NULL = bytes([0xC0]) + bytes(47)
G2_NULL = bytes([0xC0]) + bytes(95)



def _neg(group: int, enc: bytes) -> bytes:
    """-P on the ZCash compressed encoding: flip the y-sign flag (no point of the r-torsion has y = 0)."""
    if enc[0] & 0x40:  # infinity
        return enc
    return bytes([enc[0] ^ 0x20]) + enc[1:]


ENGINE = Engine("bls12381", "bls12-381", ORDER, G1_LEN, G2_LEN, GT_LEN, G1_BASE, G2_BASE, G1_NULL, G2_NULL, _neg)
g1_batch_mul, g2_batch_mul = ENGINE.g1_batch_mul, ENGINE.g2_batch_mul
g1_commit, g2_commit = ENGINE.g1_commit, ENGINE.g2_commit
batch_pair, batch_validate_pairing = ENGINE.batch_pair, ENGINE.batch_validate_pairing
_mul = ENGINE.mul
g1_msm, g2_msm = ENGINE.g1_msm, ENGINE.g2_msm
gt_batch_mul = ENGINE.gt_batch_mul
Scalar, G1Elt, G2Elt, GTElt, Suite = ENGINE.make_types()


def NewSuite() -> Suite:
    return Suite()
