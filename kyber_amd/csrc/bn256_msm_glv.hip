// pairing/bn256: the G1 MSM on balanced GLV halves (bn_msm_glv.inc).
#include "bn256.cuh"
#define KYB_BN_PFX bn256
#define KYB_BN_NS bn
#define KYB_BN_TAG Bn
#include "bn_msm_glv.inc"
