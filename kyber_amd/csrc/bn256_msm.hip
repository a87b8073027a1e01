// pairing/bn256: multi-scalar multiplication / polynomial evaluation entry points (bn_msm.inc).
#include "bn256.cuh"
#define KYB_BN_PFX bn256
#define KYB_BN_NS bn
#define KYB_BN_TAG Bn
#include "bn_msm.inc"
