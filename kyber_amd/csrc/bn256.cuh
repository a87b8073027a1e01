// pairing/bn256 (dclxvi parameters) instance of the BN device library (bn_suite.inc) + its hash to G1.
#pragma once
#include "bn256_params.h"
#include "sha256.cuh"
#define KYB_BN_NS bn
#define KYB_BN_PARAMS Bn256
#include "bn_suite.inc"
#undef KYB_BN_NS
#undef KYB_BN_PARAMS

namespace kyb {
namespace bn {
// pointG1.Hash -> hashToPoint (point.go:261-313): x = SHA-256(m) mod p; while x^3 + 3 has no square root
// x += 1; y = (x^3 + 3)^((p+1)/4) (big.Int.ModSqrt for p = 3 mod 4).  Output: 64-byte G1 encoding.
KYB_HD int hash_g1_wire(uint8_t* out, const uint8_t* msg, size_t len) {
    uint32_t h[8], w[8];
    sha256(h, msg, len);
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = h[7 - k];  // digest as a big-endian integer -> little-endian words
    fp x, y, t, b, one;
    fp_from_words<FC>(x, w);  // reduces mod p
    fp_const(b, CC::B1);
    fp_one(one);
    bool found = false;
#pragma unroll 1
    for (int iter = 0; iter < 256 && !found; iter++) {
        fp_sqr(t, x);
        fp_mul(t, t, x);
        fp_add(t, t, b);
        fp_pow_words<FC>(y, t, FC::SQRT_EXP, FC::SQRT_BITS);
        fp y2;
        fp_sqr(y2, y);
        found = fp_eq(y2, t);
        if (!found) fp_add(x, x, one);
    }
    fp_encode(out, x);
    fp_encode(out + 32, y);
    return found ? ST_OK : ST_BAD_POINT;
}
}  // namespace bn
}  // namespace kyb
