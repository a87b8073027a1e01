// Scalar-field arithmetic modulo the group order (256-bit odd modulus q, Montgomery radix 2^256) and the batched
// private-polynomial evaluation built on it:
//
//   out[i] = sum_j coeffs[j] (idx[i] + 1)^j  mod q
//
// share.PriPoly.Eval (share/poly.go:85-93: xi = 1 + i; v = v * xi + coeffs[j] from the top coefficient down) for many
// indices at once -- the loop of PriPoly.Shares (share/poly.go:96-102) that every dealer of share/vss and share/dkg
// runs once per participant.  The reference does n x t big-integer multiplications modulo q (group/mod/int.go Mul +
// Add); here a lane owns an index and walks the t coefficients, which every lane of the wave reads at the same address.
//
// Encodings are the reference's scalar encodings: Ed25519 32 bytes little-endian (group/edwards25519/scalar.go), the
// pairing suites' mod.Int 32 bytes big-endian (group/mod/int.go MarshalBinary).  Any 32-byte string is accepted as a
// coefficient and taken modulo q; outputs are canonical (< q).
#pragma once
#include "hd.h"

namespace kyb {
namespace sf {

struct Mod {
    uint32_t q[8];    // the modulus, little-endian words
    uint32_t r2[8];   // 2^512 mod q
    uint32_t qinv;    // -q^-1 mod 2^32
    uint32_t be;      // wire form: 1 = big-endian (mod.Int), 0 = little-endian (Ed25519)
};

// a >= b on 8 little-endian words (+ a ninth word of a)
KYB_HD bool geq(const uint32_t (&a)[8], uint32_t a8, const uint32_t (&b)[8]) {
    if (a8) return true;
    bool ge = true;  // equal so far, from the top
    bool decided = false;
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        const bool gt = a[i] > b[i], lt = a[i] < b[i];
        ge = decided ? ge : (gt ? true : (lt ? false : ge));
        decided = decided | gt | lt;
    }
    return ge;
}
KYB_HD void sub_q(uint32_t (&a)[8], const uint32_t (&q)[8]) {
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t d = (uint64_t)a[i] - q[i] - br;
        a[i] = (uint32_t)d;
        br = (d >> 32) & 1u;
    }
}
// r = a b / 2^256 mod q for a < 2^256, b < q (CIOS, 32-bit words, 64-bit accumulators); r < q
KYB_HD void mont_mul(uint32_t (&r)[8], const uint32_t (&a)[8], const uint32_t (&b)[8], const Mod& m) {
    uint32_t t[10];
#pragma unroll
    for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c += (uint64_t)t[j] + (uint64_t)a[j] * b[i];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[8] = (uint32_t)c;
        t[9] = (uint32_t)(c >> 32);
        const uint32_t mm = t[0] * m.qinv;
        c = (uint64_t)t[0] + (uint64_t)mm * m.q[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            c += (uint64_t)t[j] + (uint64_t)mm * m.q[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[7] = (uint32_t)c;
        t[8] = t[9] + (uint32_t)(c >> 32);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = t[i];
    if (geq(r, t[8], m.q)) sub_q(r, m.q);
}
// r = a + b mod q for a, b < q
KYB_HD void add_mod(uint32_t (&r)[8], const uint32_t (&a)[8], const uint32_t (&b)[8], const Mod& m) {
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a[i] + b[i];
        r[i] = (uint32_t)c;
        c >>= 32;
    }
    if (geq(r, (uint32_t)c, m.q)) sub_q(r, m.q);
}
KYB_HD void load(uint32_t (&w)[8], const uint8_t* p, bool be) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint8_t* s = be ? p + 28 - 4 * i : p + 4 * i;
        w[i] = be ? ((uint32_t)s[0] << 24 | (uint32_t)s[1] << 16 | (uint32_t)s[2] << 8 | s[3])
                  : ((uint32_t)s[3] << 24 | (uint32_t)s[2] << 16 | (uint32_t)s[1] << 8 | s[0]);
    }
}
KYB_HD void store(uint8_t* p, const uint32_t (&w)[8], bool be) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint8_t* s = be ? p + 28 - 4 * i : p + 4 * i;
        const uint32_t v = w[i];
        if (be) {
            s[0] = (uint8_t)(v >> 24); s[1] = (uint8_t)(v >> 16); s[2] = (uint8_t)(v >> 8); s[3] = (uint8_t)v;
        } else {
            s[0] = (uint8_t)v; s[1] = (uint8_t)(v >> 8); s[2] = (uint8_t)(v >> 16); s[3] = (uint8_t)(v >> 24);
        }
    }
}
// wire coefficient -> Montgomery residue (any 32-byte string: a < 2^256, r2 < q, so the product's bound holds)
KYB_HD void to_mont(uint32_t (&r)[8], const uint8_t* wire, const Mod& m) {
    uint32_t a[8];
    load(a, wire, m.be != 0);
    mont_mul(r, a, m.r2, m);
}
// one index: Horner over the Montgomery coefficients cm[0..t), x = idx + 1
KYB_HD void horner(uint8_t* out, uint32_t idx, size_t t, const uint32_t* cm, const Mod& m) {
    const uint64_t x = (uint64_t)idx + 1;
    uint32_t xw[8] = {(uint32_t)x, (uint32_t)(x >> 32), 0, 0, 0, 0, 0, 0}, xm[8], v[8];
    mont_mul(xm, xw, m.r2, m);
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = 0;
    for (size_t j = t; j-- > 0;) {
        uint32_t c[8], p[8];
#pragma unroll
        for (int i = 0; i < 8; i++) c[i] = cm[8 * j + i];
        mont_mul(p, v, xm, m);
        add_mod(v, p, c, m);
    }
    const uint32_t one[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    uint32_t r[8];
    mont_mul(r, v, one, m);  // leaves the Montgomery form; canonical
    store(out, r, m.be != 0);
}

// host side: the derived constants of a modulus (2^512 mod q by 512 modular doublings, -q^-1 by Newton's iteration)
inline Mod make_mod(const uint32_t (&q)[8], bool be) {
    Mod m;
    for (int i = 0; i < 8; i++) m.q[i] = q[i];
    m.be = be ? 1u : 0u;
    uint32_t inv = 1;
    for (int i = 0; i < 5; i++) inv *= 2u - q[0] * inv;  // q^-1 mod 2^32
    m.qinv = 0u - inv;
    uint32_t r[8] = {1, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < 512; k++) {
        uint64_t c = 0;
        for (int i = 0; i < 8; i++) {
            c += (uint64_t)r[i] * 2u;
            r[i] = (uint32_t)c;
            c >>= 32;
        }
        if (geq(r, (uint32_t)c, m.q)) sub_q(r, m.q);
    }
    for (int i = 0; i < 8; i++) m.r2[i] = r[i];
    return m;
}

// The four group orders (little-endian words).  tests/test_constants.py holds them against the oracles' ORDER values.
//   Ed25519: l = 2^252 + 27742317777372353535851937790883648493 (group/edwards25519/const.go primeOrder)
//   BLS12-381: r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
//   bn256: n = 65000549695646603732796438742359905742570406053903786389881062969044166799969 (pairing/bn256/constants.go Order)
//   bn254 (alt_bn128): n = 21888242871839275222246405745257275088548364400416034343698204186575808495617
constexpr uint32_t Q_ED25519[8] = {0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0, 0, 0, 0x10000000u};
constexpr uint32_t Q_BLS12381[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
constexpr uint32_t Q_BN256[8] = {0x57ac7261u, 0x1a2ef45bu, 0xf82b3924u, 0x2e8d8e12u, 0x6184dc21u, 0xaa6fecb8u, 0x4aa387f9u, 0x8fb501e3u};
constexpr uint32_t Q_BN254[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};

}  // namespace sf
}  // namespace kyb
