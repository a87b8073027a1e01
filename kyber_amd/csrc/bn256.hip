// bn256 batch kernels for gfx950 + their C-ABI entry points (stamped out by pairing_abi.cuh).
//
// Replaces pairing/bn256 (in-tree arithmetic):
//   pointG1.Mul / pointG2.Mul        point.go:154,405 -> curve.go:189 / twist.go:162  -> bn256_g1_mul_kernel / _g2_mul_kernel
//   Suite.Pair                       suite.go:97 -> optate.go:266                     -> bn256_pair_kernel
//   Suite.ValidatePairing            suite.go:105-107 (two pairings + Equal)          -> bn256_pair_check_kernel
//   (Un)MarshalBinary                point.go:170-238, 423-499, 630-662               -> fused into every kernel
// (this translation unit: G1 / G2 scalar multiplication; pairing kernels are in bn256_pair.hip, MSM in
//  bn256_msm.hip -- split only so that the three compile in parallel)
#include "bn256.cuh"
#include "pairing_abi.cuh"

KYB_DEFINE_MUL_ABI(bn256, bn, 64, 128)
