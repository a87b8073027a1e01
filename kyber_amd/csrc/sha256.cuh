// SHA-256 (FIPS 180-4), one message per lane: one-shot digest for the bn256 hash-to-point kernel
// (pairing/bn256/point.go:286-288) and an incremental context for expand_message_xmd (RFC 9380 section 5.3.1,
// used by the BLS12-381 hash-to-curve kernels).  All lanes of a launch hash inputs of the same length, so
// control flow is uniform.
#pragma once
#include "hd.h"

namespace kyb {

KYB_HD uint32_t sha_rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

KYB_HD_NOINLINE void sha256_block(uint32_t (&h)[8], const uint32_t (&blk)[16]) {
    constexpr uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
        0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
        0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
        0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
        0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
        0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t w[64];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = blk[i];
#pragma unroll
    for (int i = 16; i < 64; i++) {
        const uint32_t s0 = sha_rotr(w[i - 15], 7) ^ sha_rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
        const uint32_t s1 = sha_rotr(w[i - 2], 17) ^ sha_rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        const uint32_t S1 = sha_rotr(e, 6) ^ sha_rotr(e, 11) ^ sha_rotr(e, 25);
        const uint32_t ch = (e & f) ^ (~e & g);
        const uint32_t t1 = hh + S1 + ch + K[i] + w[i];
        const uint32_t S0 = sha_rotr(a, 2) ^ sha_rotr(a, 13) ^ sha_rotr(a, 22);
        const uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        const uint32_t t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

// Incremental context: absorb bytes, then finish() leaves the digest in h (eight big-endian words, h[0] first).
struct Sha256 {
    uint32_t h[8];
    uint32_t w[16];   // current block, big-endian words being filled
    uint64_t len;     // bytes absorbed
    KYB_HD void init() {
        h[0] = 0x6a09e667; h[1] = 0xbb67ae85; h[2] = 0x3c6ef372; h[3] = 0xa54ff53a;
        h[4] = 0x510e527f; h[5] = 0x9b05688c; h[6] = 0x1f83d9ab; h[7] = 0x5be0cd19;
        len = 0;
        for (int i = 0; i < 16; i++) w[i] = 0;
    }
    KYB_HD void put(uint8_t byte) {
        const int pos = (int)(len & 63);
        const int wi = pos >> 2, sh = 24 - 8 * (pos & 3);
        w[wi] |= (uint32_t)byte << sh;
        len++;
        if ((len & 63) == 0) {
            sha256_block(h, w);
            for (int i = 0; i < 16; i++) w[i] = 0;
        }
    }
    KYB_HD void update(const uint8_t* p, size_t n) {
        for (size_t i = 0; i < n; i++) put(p[i]);
    }
    KYB_HD void update_words_be(const uint32_t* d, int nwords) {  // nwords big-endian 32-bit words
        for (int i = 0; i < nwords; i++)
            for (int k = 0; k < 4; k++) put((uint8_t)(d[i] >> (24 - 8 * k)));
    }
    KYB_HD void finish() {
        const uint64_t bits = len * 8;
        put(0x80);
        while ((len & 63) != 56) put(0);
        w[14] = (uint32_t)(bits >> 32);
        w[15] = (uint32_t)bits;
        sha256_block(h, w);
    }
};

// digest (eight big-endian words, h[0] first) of msg[0..len)
KYB_HD_NOINLINE void sha256(uint32_t (&h)[8], const uint8_t* msg, size_t len) {
    Sha256 c;
    c.init();
    c.update(msg, len);
    c.finish();
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = c.h[i];
}

}  // namespace kyb
