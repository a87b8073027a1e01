// pairing/bn254: multi-scalar multiplication / polynomial evaluation entry points (bn_msm.inc).
#include "bn254.cuh"
#define KYB_BN_PFX bn254
#define KYB_BN_NS bn4
#define KYB_BN_TAG Bn4
#include "bn_msm.inc"
