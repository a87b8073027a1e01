// BLS12-381 batch kernels for gfx950 + their C-ABI entry points (stamped out by pairing_abi.cuh):
// one group operation / pairing per lane, integer VALU only, uniform control flow inside a wave
// except for rejected inputs.
//
// Replaces, behind pairing/bls12381/kilic (the adapter whose Point()/Pair() the suite exposes):
//   G1Elt.Mul / G2Elt.Mul            kilic/g1.go:110-116, g2.go  -> bls12381_g1_mul_kernel / _g2_mul_kernel
//   Suite.Pair                       kilic/suite.go:70-75        -> bls12381_pair_kernel
//   Suite.ValidatePairing            kilic/suite.go:57-68        -> bls12381_pair_check_kernel
//   Unmarshal/MarshalBinary          kilic/g1.go:119-131         -> fused into every kernel
// (this translation unit: G1 / G2 scalar multiplication; pairing kernels are in bls12381_pair.hip, MSM in
//  bls12381_msm.hip -- split only so that the three compile in parallel)
#include "bls12381.cuh"
#include "pairing_abi.cuh"

KYB_DEFINE_MUL_ABI(bls12381, bls, 48, 96)
