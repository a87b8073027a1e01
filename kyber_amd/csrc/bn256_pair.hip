// bn256: pairing, pairing-check and GT exponentiation kernels + C-ABI entry points (see bn256.hip for
// the map to the reference functions they replace).
#include "bn256.cuh"
#include "pairing_abi.cuh"

KYB_DEFINE_PAIR_ABI(bn256, bn, 64, 128, 384)
