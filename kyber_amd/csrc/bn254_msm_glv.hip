// pairing/bn254: the G1 MSM on balanced GLV halves (bn_msm_glv.inc).
#include "bn254.cuh"
#define KYB_BN_PFX bn254
#define KYB_BN_NS bn4
#define KYB_BN_TAG Bn4
#include "bn_msm_glv.inc"
