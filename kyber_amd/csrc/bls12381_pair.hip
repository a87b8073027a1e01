// bls12381: pairing, pairing-check and GT exponentiation kernels + C-ABI entry points (see bls12381.hip for
// the map to the reference functions they replace).
#include "bls12381.cuh"
#include "pairing_abi.cuh"

KYB_DEFINE_PAIR_ABI(bls12381, bls, 48, 96, 576)
